// SURVEY 8f row 1, the memory-bound tail of MarginalNet.forward (MN/model.py:73-88) around the library GEMMs:
//   masked_token_mean : t = (txt_tok * txt_mask).sum(1) / txt_mask.sum(1).clamp_min(1)         model.py:77-78
//   film_axis_means   : v = gamma*v + beta; vx = v.mean(2); vy = v.mean(3)                     model.py:80-88
// The reference runs these as 3 + 4 elementwise / reduction kernels that each stream the whole tensor; here each
// tensor is read once.  The convolutions and linear layers between them stay on the stock library (MFMA-class
// GEMMs, out of the hand-kernel scope).
//
// Arithmetic (mirrored by oracle/warp_oracle.py::masked_token_mean / film_axis_means): every elementwise
// product / sum is one float32 rounding as in the reference's separate kernels; reductions whose order torch
// leaves implementation defined are accumulated in float64 and rounded once.
#include "common.hpp"

namespace attwarp {
namespace mnet {

constexpr int NT = 256;

// grid = (ceil(D/NT), B); thread d streams tok[b, :, d] (consecutive lanes -> consecutive addresses)
template <typename T>
__global__ __launch_bounds__(NT) void masked_token_mean_kernel(const T* __restrict__ tok,
                                                               const float* __restrict__ mask, int Lt, int D,
                                                               float* __restrict__ out) {
  const int b = blockIdx.y;
  const int d = blockIdx.x * NT + threadIdx.x;
  const float* m = mask + (size_t)b * Lt;
  double msum = 0.0;
  for (int l = 0; l < Lt; ++l) msum += (double)m[l];
  if (d >= D) return;
  const T* tp = tok + (size_t)b * Lt * D + d;
  double acc = 0.0;
  int l = 0;
  for (; l + 4 <= Lt; l += 4) {          // four independent loads in flight
    const float a0 = to_f32<T>(tp[(size_t)(l + 0) * D]), a1 = to_f32<T>(tp[(size_t)(l + 1) * D]);
    const float a2 = to_f32<T>(tp[(size_t)(l + 2) * D]), a3 = to_f32<T>(tp[(size_t)(l + 3) * D]);
    acc += (double)fmul(a0, m[l + 0]);
    acc += (double)fmul(a1, m[l + 1]);
    acc += (double)fmul(a2, m[l + 2]);
    acc += (double)fmul(a3, m[l + 3]);
  }
  for (; l < Lt; ++l) acc += (double)fmul(to_f32<T>(tp[(size_t)l * D]), m[l]);
  const float denom = fmaxf((float)msum, 1.0f);
  out[(size_t)b * D + d] = (float)acc / denom;
}

// One wave per (b, channel) plane: FiLM the plane into an LDS tile (rows padded by one float), then lane t sums
// column t (t < W) or row t - W.  grid = ceil(B*Ch / 4), 4 planes per workgroup.
__global__ __launch_bounds__(NT) void film_axis_means_kernel(const float* __restrict__ v,
                                                             const float* __restrict__ gamma_beta, int planes, int Ch,
                                                             int H, int W, float* __restrict__ vx,
                                                             float* __restrict__ vy) {
  extern __shared__ float tiles[];
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  const int WP = W + 1;
  float* tile = tiles + (size_t)wid * H * WP;
  const int pl = blockIdx.x * (NT / WAVE) + wid;
  const bool live = pl < planes;
  if (live) {
    const int b = pl / Ch, c = pl - b * Ch;
    const float gamma = gamma_beta[(size_t)b * 2 * Ch + c], beta = gamma_beta[(size_t)b * 2 * Ch + Ch + c];
    const float* src = v + (size_t)pl * H * W;
    for (int e = lane; e < H * W; e += WAVE) {
      const int y = e / W, x = e - y * W;
      tile[y * WP + x] = fadd(fmul(gamma, src[e]), beta);
    }
  }
  __syncthreads();
  if (!live) return;
  for (int t = lane; t < W + H; t += WAVE) {
    double acc = 0.0;
    if (t < W) {
      for (int y = 0; y < H; ++y) acc += (double)tile[y * WP + t];
      vx[(size_t)pl * W + t] = (float)acc / (float)H;        // mean over Y
    } else {
      const float* row = tile + (t - W) * WP;
      for (int x = 0; x < W; ++x) acc += (double)row[x];
      vy[(size_t)pl * H + (t - W)] = (float)acc / (float)W;  // mean over X
    }
  }
}

}  // namespace mnet
}  // namespace attwarp

using namespace attwarp;

extern "C" int attwarp_masked_token_mean(const void* tok, int dtype, const float* mask, int B, int Lt, int D,
                                         float* out, void* stream) {
  ATTWARP_REQUIRE(tok && mask && out, "masked_token_mean: null pointer");
  ATTWARP_REQUIRE(B > 0 && Lt > 0 && D > 0, "masked_token_mean: non-positive size");
  ATTWARP_REQUIRE(dtype == ATTWARP_F32 || dtype == ATTWARP_F16 || dtype == ATTWARP_BF16,
                  "masked_token_mean: dtype must be F32, F16 or BF16 (got %d)", dtype);
  if (B > 65535) return fail(ATTWARP_E_UNSUPPORTED, "masked_token_mean: B=%d > 65535", B);
  const dim3 grid((D + mnet::NT - 1) / mnet::NT, B);
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case ATTWARP_F32:
      hipLaunchKernelGGL((mnet::masked_token_mean_kernel<float>), grid, dim3(mnet::NT), 0, st, (const float*)tok, mask, Lt,
                         D, out);
      break;
    case ATTWARP_F16:
      hipLaunchKernelGGL((mnet::masked_token_mean_kernel<__half>), grid, dim3(mnet::NT), 0, st, (const __half*)tok, mask,
                         Lt, D, out);
      break;
    default:
      hipLaunchKernelGGL((mnet::masked_token_mean_kernel<__hip_bfloat16>), grid, dim3(mnet::NT), 0, st,
                         (const __hip_bfloat16*)tok, mask, Lt, D, out);
      break;
  }
  return check_launch("masked_token_mean_kernel");
}

extern "C" int attwarp_film_axis_means(const float* v, const float* gamma_beta, int B, int Ch, int H, int W, float* vx,
                                       float* vy, void* stream) {
  ATTWARP_REQUIRE(v && gamma_beta && vx && vy, "film_axis_means: null pointer");
  ATTWARP_REQUIRE(B > 0 && Ch > 0 && H > 0 && W > 0, "film_axis_means: non-positive size");
  const long long tile = (long long)H * (W + 1);
  if (tile > 4096) return fail(ATTWARP_E_UNSUPPORTED, "film_axis_means: H*(W+1)=%lld > 4096", tile);
  const long long planes = (long long)B * Ch;
  if (planes > 2147483647LL) return fail(ATTWARP_E_UNSUPPORTED, "film_axis_means: B*Ch too large");
  const int per = mnet::NT / WAVE;
  const size_t lds = (size_t)per * tile * sizeof(float);
  hipLaunchKernelGGL(mnet::film_axis_means_kernel, dim3((unsigned)((planes + per - 1) / per)), dim3(mnet::NT), lds,
                     as_stream(stream), v, gamma_beta, (int)planes, Ch, H, W, vx, vy);
  return check_launch("film_axis_means_kernel");
}
