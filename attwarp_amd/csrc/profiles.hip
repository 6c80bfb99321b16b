// Stages that read a full-resolution attention map once:
//   A5  F.adaptive_avg_pool2d(A,(24,24))                    MN/trainer.py:197,433,465
//   A6  gt_marginals                                        MN/checkpoint_utils.py:43-51
//   A13 marginals -> CDF -> inverse map (float64 variant)   AGW/new_method.py:206-265
//
// All three are one coalesced pass over [B,H,W] (HBM-bound, S*S*elemsize algorithmic bytes per
// image -- SURVEY 8d) followed by per-axis work on H+W numbers.  Sums are accumulated in double;
// column sums are accumulated row by row in ascending order, which is the order numpy uses for
// np.sum(axis=0) on a C-contiguous array, so the x profile of A13 is bit-identical to numpy's.
#include "common.hpp"
#include "interp.hpp"

namespace attwarp {

constexpr int NT = 256;

// ---- element transform applied while summing -------------------------------------------
struct XfClampPos {  // gt_marginals: A.clamp_min(0)
  __device__ __forceinline__ double operator()(double v) const { return (v != v) ? v : (v > 0.0 ? v : 0.0); }  // NaN propagates
};
struct XfAttention {  // new_method: max(att,0) -> transform -> + BASE_ATTENTION
  int transform;
  double exp_scale, exp_divisor;
  __device__ __forceinline__ double operator()(double v) const {
    double a = (v != v) ? v : (v > 0.0 ? v : 0.0);       // np.maximum(x, 0) propagates NaN
    switch (transform) {
      case ATTWARP_T_SQUARE: a = a * a; break;
      case ATTWARP_T_SQRT: a = sqrt((a != a) ? a : (a > 0.0 ? a : 0.0)); break;
      case ATTWARP_T_EXP: a = exp(exp_scale * a) / exp_divisor; break;
      case ATTWARP_T_LOG: a = log(a + 1e-5); break;
      default: break;
    }
    return a + 1e-9;
  }
};

template <typename T> __device__ __forceinline__ double ld_f64(const T* p) { return (double)*p; }
template <> __device__ __forceinline__ double ld_f64<uint8_t>(const uint8_t* p) { return (double)(int)*p; }

// Column sums: one thread per column, rows in ascending order.  grid = (ceil(W/NT), B)
template <typename T, typename XF>
__global__ __launch_bounds__(NT) void col_sums_kernel(const T* __restrict__ A, int H, int W, XF xf,
                                                      double* __restrict__ col) {
  const int b = blockIdx.y;
  const int x = blockIdx.x * NT + threadIdx.x;
  if (x >= W) return;
  const T* base = A + (size_t)b * H * W + x;
  double acc = 0.0;
  int r = 0;
  for (; r + 8 <= H; r += 8) {
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = xf(ld_f64(base + (size_t)(r + i) * W));
#pragma unroll
    for (int i = 0; i < 8; ++i) acc = acc + v[i];
  }
  for (; r < H; ++r) acc = acc + xf(ld_f64(base + (size_t)r * W));
  col[(size_t)b * W + x] = acc;
}

// Row sums: one wave per row.  grid = (ceil(H/4), B)
template <typename T, typename XF>
__global__ __launch_bounds__(NT) void row_sums_kernel(const T* __restrict__ A, int H, int W, XF xf,
                                                      double* __restrict__ row) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & (WAVE - 1);
  const int y = blockIdx.x * (NT / WAVE) + threadIdx.x / WAVE;
  if (y >= H) return;
  const T* base = A + ((size_t)b * H + y) * W;
  double acc = 0.0;
  for (int x = lane; x < W; x += WAVE) acc = acc + xf(ld_f64(base + x));
  acc = wave_sum(acc);
  if (lane == 0) row[(size_t)b * H + y] = acc;
}

// ---- A6 finalize: normalise a marginal.  grid = (B, 2) ----------------------------------
__global__ __launch_bounds__(NT) void marginals_finalize_kernel(const double* __restrict__ col,
                                                                const double* __restrict__ row, int H, int W,
                                                                float* __restrict__ px, float* __restrict__ py) {
  __shared__ double red[NT / WAVE];
  const int b = blockIdx.x, axis = blockIdx.y;
  const int n = axis ? H : W;
  const double* src = (axis ? row : col) + (size_t)b * n;
  float* dst = (axis ? py : px) + (size_t)b * n;
  double acc = 0.0;
  for (int k = threadIdx.x; k < n; k += blockDim.x) acc += (double)(float)src[k];
  const float tot = fmaxf((float)block_sum(acc, red), 1e-6f);
  for (int k = threadIdx.x; k < n; k += blockDim.x) dst[k] = (float)src[k] / tot;
}

// ---- A5: adaptive average pool.  grid = (oh, B), one band of rows per block ---------------
__global__ __launch_bounds__(NT) void adaptive_pool_kernel(const float* __restrict__ A, int H, int W, int oh, int ow,
                                                           float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) double colsum[];   // W doubles
  const int b = blockIdx.y, i = blockIdx.x;
  const int ys = (int)(((long long)i * H) / oh);
  const int ye = (int)((((long long)(i + 1)) * H + oh - 1) / oh);
  const float* base = A + (size_t)b * H * W;
  for (int x = threadIdx.x; x < W; x += blockDim.x) {
    double acc = 0.0;
    int r = ys;
    for (; r + 4 <= ye; r += 4) {
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = base[(size_t)(r + q) * W + x];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc += (double)v[q];
    }
    for (; r < ye; ++r) acc += (double)base[(size_t)r * W + x];
    colsum[x] = acc;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < ow; j += blockDim.x) {
    const int xs = (int)(((long long)j * W) / ow);
    const int xe = (int)((((long long)(j + 1)) * W + ow - 1) / ow);
    double acc = 0.0;
    for (int x = xs; x < xe; ++x) acc += colsum[x];
    const float cnt = (float)((ye - ys) * (xe - xs));
    out[((size_t)b * oh + i) * ow + j] = (float)acc / cnt;
  }
}

// ---- A13 finalize: profile -> (inverse) -> total / fallback -> cumsum -> knots -> np.interp ----
// AGW/new_method.py:218-261.  grid = (B, 2); LDS: (n+1) doubles.
__device__ __forceinline__ double inverse_transform(double x, int transform, double exp_scale, double exp_divisor) {
  switch (transform) {
    case ATTWARP_T_SQUARE: return sqrt((x != x) ? x : (x > 0.0 ? x : 0.0));
    case ATTWARP_T_SQRT: return x * x;
    case ATTWARP_T_EXP: {
      const double t = x * exp_divisor;
      return log((t != t) ? t : (t > 1e-9 ? t : 1e-9)) / exp_scale;
    }
    case ATTWARP_T_LOG: return exp(x) - 1e-5;
    default: return x;
  }
}

__global__ __launch_bounds__(NT) void attention_maps_finalize_kernel(const double* __restrict__ col,
                                                                     const double* __restrict__ row, int h, int w,
                                                                     int new_w, int new_h, int transform,
                                                                     double exp_scale, double exp_divisor,
                                                                     int apply_inverse, float* __restrict__ map_x,
                                                                     float* __restrict__ map_y) {
  extern __shared__ __attribute__((aligned(16))) double smem_d[];
  __shared__ double red[NT / WAVE];
  const int b = blockIdx.x, axis = blockIdx.y;
  const int n = axis ? h : w;            // profile length
  const int other = axis ? w : h;        // number of terms summed into each profile entry
  const int n_out = axis ? new_h : new_w;
  const double* prof_in = (axis ? row : col) + (size_t)b * n;
  float* map = (axis ? map_y : map_x) + (size_t)b * n_out;
  double* xn = smem_d;                   // n+1 knots; xn[1..n] first holds the profile

  // sum of all biased values (for the fallback's np.mean): use the row profile before any inverse
  double all = 0.0;
  {
    const double* r = row + (size_t)b * h;
    for (int k = threadIdx.x; k < h; k += blockDim.x) all += r[k];
    all = block_sum(all, red);
  }
  double acc = 0.0, acc_other = 0.0;
  for (int k = threadIdx.x; k < n; k += blockDim.x) {
    double v = prof_in[k];
    if (apply_inverse) {
      v = inverse_transform(v - 1e-9 * (double)other, transform, exp_scale, exp_divisor);
      v = v + 1e-9 * (double)other;
    }
    xn[k + 1] = v;
    acc += v;
  }
  const double total_self = block_sum(acc, red);
  {
    // the fallback test looks at BOTH totals (total_att_x < EPS or total_att_y < EPS)
    const int m = axis ? w : h;
    const double* o = (axis ? col : row) + (size_t)b * m;
    for (int k = threadIdx.x; k < m; k += blockDim.x) {
      double v = o[k];
      if (apply_inverse) {
        v = inverse_transform(v - 1e-9 * (double)n, transform, exp_scale, exp_divisor);
        v = v + 1e-9 * (double)n;
      }
      acc_other += v;
    }
    acc_other = block_sum(acc_other, red);
  }
  double total = total_self;
  const bool fallback = (total_self < 1e-9) || (acc_other < 1e-9);
  __syncthreads();
  if (fallback) {
    for (int k = threadIdx.x; k < n; k += blockDim.x) xn[k + 1] = 1.0;
    // total_att_x = w * (np.mean(att_map_biased) * h); total_att_y = h * (mean * w); then max(., EPS)
    const double mean = all / ((double)h * (double)w);
    total = (double)n * (mean * (double)other);
    total = (total != total) ? total : (total > 1e-9 ? total : 1e-9);   // python max(nan-first?) see note
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    // np.cumsum (sequential), / total, * new ; knot 0 = 0 * new ; last knot = new
    double c = 0.0;
    for (int k = 1; k <= n; ++k) {
      c = c + xn[k];
      xn[k] = (c / total) * (double)n_out;
    }
    xn[0] = 0.0;
    xn[n] = (double)n_out;
  }
  __syncthreads();
  const bool mono = block_is_sorted(xn, n + 1);
  np_interp_block(xn, n + 1, n_out, map, mono);
}

template <typename T, typename XF>
static int launch_axis_sums(const void* A, int B, int H, int W, XF xf, double* col, double* row, hipStream_t st) {
  hipLaunchKernelGGL((col_sums_kernel<T, XF>), dim3((W + NT - 1) / NT, B), dim3(NT), 0, st, (const T*)A, H, W, xf, col);
  int rc = check_launch("col_sums_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL((row_sums_kernel<T, XF>), dim3((H + 3) / 4, B), dim3(NT), 0, st, (const T*)A, H, W, xf, row);
  return check_launch("row_sums_kernel");
}

}  // namespace attwarp

using namespace attwarp;

extern "C" size_t attwarp_axis_sums_workspace_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return (size_t)B * ((size_t)H + (size_t)W) * sizeof(double);
}

extern "C" int attwarp_gt_marginals(const float* A, int B, int H, int W, float* px, float* py, void* ws,
                                    void* stream) {
  ATTWARP_REQUIRE(A && px && py && ws, "gt_marginals: null pointer");
  ATTWARP_REQUIRE(B > 0 && H > 0 && W > 0, "gt_marginals: non-positive size");
  if (B > 65535) return fail(ATTWARP_E_UNSUPPORTED, "gt_marginals: B > 65535");
  double* col = (double*)ws;
  double* row = col + (size_t)B * W;
  hipStream_t st = as_stream(stream);
  int rc = launch_axis_sums<float, XfClampPos>(A, B, H, W, XfClampPos{}, col, row, st);
  if (rc) return rc;
  hipLaunchKernelGGL(marginals_finalize_kernel, dim3(B, 2), dim3(NT), 0, st, col, row, H, W, px, py);
  return check_launch("marginals_finalize_kernel");
}

extern "C" int attwarp_adaptive_avg_pool(const float* A, int B, int H, int W, int oh, int ow, float* out,
                                         void* stream) {
  ATTWARP_REQUIRE(A && out, "adaptive_avg_pool: null pointer");
  ATTWARP_REQUIRE(B > 0 && H > 0 && W > 0 && oh > 0 && ow > 0, "adaptive_avg_pool: non-positive size");
  if (oh > 65535 || B > 65535) return fail(ATTWARP_E_UNSUPPORTED, "adaptive_avg_pool: oh/B > 65535");
  if (W > 16384) return fail(ATTWARP_E_UNSUPPORTED, "adaptive_avg_pool: W=%d > 16384", W);
  hipLaunchKernelGGL(adaptive_pool_kernel, dim3(oh, B), dim3(NT), (size_t)W * sizeof(double), as_stream(stream), A, H,
                     W, oh, ow, out);
  return check_launch("adaptive_pool_kernel");
}

extern "C" int attwarp_axis_maps_from_attention(const void* att, int dtype, int B, int h, int w, int new_w, int new_h,
                                                int transform, double exp_scale, double exp_divisor,
                                                int apply_inverse, float* map_x, float* map_y, void* ws,
                                                void* stream) {
  ATTWARP_REQUIRE(att && map_x && map_y && ws, "axis_maps_from_attention: null pointer");
  ATTWARP_REQUIRE(B > 0 && h > 0 && w > 0 && new_w > 0 && new_h > 0, "axis_maps_from_attention: non-positive size");
  ATTWARP_REQUIRE(transform >= ATTWARP_T_IDENTITY && transform <= ATTWARP_T_LOG,
                  "axis_maps_from_attention: unknown transform %d", transform);
  if (B > 65535) return fail(ATTWARP_E_UNSUPPORTED, "axis_maps_from_attention: B > 65535");
  const int n = h > w ? h : w;
  if (n > 16384) return fail(ATTWARP_E_UNSUPPORTED, "axis_maps_from_attention: max(h,w)=%d > 16384", n);
  double* col = (double*)ws;
  double* row = col + (size_t)B * w;
  hipStream_t st = as_stream(stream);
  XfAttention xf{transform, exp_scale, exp_divisor};
  int rc;
  switch (dtype) {
    case ATTWARP_U8: rc = launch_axis_sums<uint8_t, XfAttention>(att, B, h, w, xf, col, row, st); break;
    case ATTWARP_F32: rc = launch_axis_sums<float, XfAttention>(att, B, h, w, xf, col, row, st); break;
    case ATTWARP_F64: rc = launch_axis_sums<double, XfAttention>(att, B, h, w, xf, col, row, st); break;
    default: return fail(ATTWARP_E_ARG, "axis_maps_from_attention: dtype must be U8, F32 or F64 (got %d)", dtype);
  }
  if (rc) return rc;
  hipLaunchKernelGGL(attention_maps_finalize_kernel, dim3(B, 2), dim3(NT), (size_t)(n + 2) * sizeof(double), st, col,
                     row, h, w, new_w, new_h, transform, exp_scale, exp_divisor, apply_inverse, map_x, map_y);
  return check_launch("attention_maps_finalize_kernel");
}
