// Per-axis 1-D stages: A7 safe_softmax, A8 right-inverse PDF up-sample, A9 PDF->CDF,
// A10 CDF repair / resample, A11 inverse map (np.interp) -- and the A8+A9+A11 fusion.
//
// Each (sample, axis) is at most a few thousand elements: one 256-thread workgroup per row, data
// staged in LDS, float64 wherever the reference's libraries accumulate or interpolate in double
// (torch-CPU cumsum, np.interp, np.concatenate promoting to float64).  These kernels are latency
// class (microseconds); the HBM-bound kernel of the path is remap_rows_kernel.
//
// Scans are done SEQUENTIALLY by one lane on purpose: the reference's cumulative sums are
// sequential double accumulations (torch cumsum on CPU, np.cumsum) and a tree scan would round
// differently; 1024 dependent double adds cost a few microseconds per workgroup, all (sample,
// axis) pairs run concurrently on different CUs.
#include "common.hpp"
#include "interp.hpp"
#include "axis_blocks.hpp"

namespace attwarp {

constexpr int NT = 256;
constexpr int MAX_L = 16384;  // LDS: (L+1) doubles + L floats

__global__ __launch_bounds__(NT) void axis_map_from_cdf_kernel(const float* __restrict__ F, int L, int n_out,
                                                               float* __restrict__ map) {
  extern __shared__ __attribute__((aligned(16))) double smem_d[];
  const int b = blockIdx.x;
  map_from_cdf_block(F + (size_t)b * L, L, n_out, smem_d, map + (size_t)b * n_out);
}

__global__ __launch_bounds__(NT) void cdf_from_density_kernel(const float* __restrict__ pin, int L,
                                                              float* __restrict__ Fout) {
  extern __shared__ __attribute__((aligned(16))) double smem_d[];
  double* red = smem_d;                                   // 8 doubles
  float* p = reinterpret_cast<float*>(smem_d + 8);
  const int b = blockIdx.x;
  for (int k = threadIdx.x; k < L; k += blockDim.x) p[k] = pin[(size_t)b * L + k];
  __syncthreads();
  cdf_from_density_block(p, L, red);
  for (int k = threadIdx.x; k < L; k += blockDim.x) Fout[(size_t)b * L + k] = p[k];
}

__global__ __launch_bounds__(NT) void right_inverse_kernel(const float* __restrict__ y, int Lo, int L,
                                                           const double* __restrict__ inv, float* __restrict__ out) {
  __shared__ float tmp[64];
  const int n = blockIdx.x;
  right_inverse_block(y + (size_t)n * Lo, Lo, L, inv, tmp, out + (size_t)n * L, false);
}

// ---- A8 + clamp_min(0) + A9 + A11 fused: the inference chain MN/trainer.py:285-289 ------
// grid = (B, 2): y = 0 -> x axis, y = 1 -> y axis.
__global__ __launch_bounds__(NT) void axis_maps_from_pdf_kernel(const float* __restrict__ px,
                                                                const float* __restrict__ py, int Lo, int W, int H,
                                                                int W_out, int H_out,
                                                                const double* __restrict__ inv_x,
                                                                const double* __restrict__ inv_y,
                                                                float* __restrict__ map_x, float* __restrict__ map_y) {
  extern __shared__ __attribute__((aligned(16))) double smem_d[];
  __shared__ float tmp[64];
  const int b = blockIdx.x, axis = blockIdx.y;
  const int L = axis ? H : W, n_out = axis ? H_out : W_out;
  const float* y = (axis ? py : px) + (size_t)b * Lo;
  const double* inv = axis ? inv_y : inv_x;
  float* map = (axis ? map_y : map_x) + (size_t)b * n_out;
  double* red = smem_d;                         // 8
  double* xn = smem_d + 8;                      // L+1
  float* p = reinterpret_cast<float*>(xn + L + 1 + ((L + 1) & 1));   // L floats
  right_inverse_block(y, Lo, L, inv, tmp, p, true);
  cdf_from_density_block(p, L, red);
  map_from_cdf_block(p, L, n_out, xn, map);
}

// ---- A2 + A6 + A8 + A9 + A11 fused: per-step attention maps -> inverse maps, one launch (body: axis_blocks.hpp) ------
// grid = (B, 2): y = 0 -> x axis, y = 1 -> y axis.
template <typename ST>
__global__ __launch_bounds__(NT) void axis_maps_from_steps_kernel(const StepsMapsArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem_d[];
  __shared__ float tmp[64];
  __shared__ float pm[64];
  axis_maps_from_steps_block<24, ST>(a, blockIdx.x, blockIdx.y, smem_d, tmp, pm);
}

// ---- A7: safe_softmax over dim=1, MN/model.py:8-14 ---------------------------------------
__global__ __launch_bounds__(NT) void safe_softmax_kernel(const float* __restrict__ logits, int N, float eps,
                                                          float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) double smem_d[];
  double* red = smem_d;
  __shared__ float wred[NT / WAVE];
  float* x = reinterpret_cast<float*>(smem_d + 8);
  const int b = blockIdx.x;
  float mx = -INFINITY;
  for (int k = threadIdx.x; k < N; k += blockDim.x) {
    float v = logits[(size_t)b * N + k];
    v = (isnan(v) || isinf(v)) ? 0.0f : v;      // nan_to_num(nan=0, posinf=0, neginf=0)
    x[k] = v;
    mx = fmaxf(mx, v);
  }
  mx = wave_max(mx);
  if ((threadIdx.x & (WAVE - 1)) == 0) wred[threadIdx.x / WAVE] = mx;
  __syncthreads();
  mx = wred[0];
  for (int i = 1; i < NT / WAVE; ++i) mx = fmaxf(mx, wred[i]);
  // logits - amax; F.softmax subtracts its own max (0 now) and exponentiates
  double acc = 0.0;
  for (int k = threadIdx.x; k < N; k += blockDim.x) {
    const float z = fsub(fsub(x[k], mx), 0.0f);
    const float e = (float)exp((double)z);
    x[k] = e;
    acc += (double)e;
  }
  const float s = (float)block_sum(acc, red);
  double acc2 = 0.0;
  for (int k = threadIdx.x; k < N; k += blockDim.x) {
    float pv = x[k] / s;
    pv = (isnan(pv) || isinf(pv)) ? 0.0f : pv;
    x[k] = pv;
    acc2 += (double)pv;
  }
  const float d = fmaxf((float)block_sum(acc2, red), eps);
  for (int k = threadIdx.x; k < N; k += blockDim.x) out[(size_t)b * N + k] = x[k] / d;
}

// ---- A10: _make_strictly_increasing (MN/checkpoint_utils.py:17-28), in LDS -----------------
// f: LDS float[N] in/out.
__device__ void strictly_increasing_block(float* f, int N, double eps) {
  for (int k = threadIdx.x; k < N; k += blockDim.x) {
    float v = f[k];
    if (isnan(v)) v = 0.0f; else if (isinf(v)) v = v > 0 ? 1.0f : 0.0f;   // nan_to_num(0, 1, 0)
    f[k] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // cummax, then diffs clamped to min_step, re-accumulated in double (torch cumsum), all sequential
    const float min_step = (float)(eps / (double)max(N, 1));
    float run = f[0];
    const float first = run;
    double c = 0.0;
    float prev_nd = run;
    for (int k = 1; k < N; ++k) {
      run = fmaxf(run, f[k]);
      const float d = fmaxf(fsub(run, prev_nd), min_step);
      prev_nd = run;
      c += (double)d;
      f[k] = fadd(first, (float)c);
    }
    const float last = fmaxf(f[N - 1], 1e-6f);
    for (int k = 0; k < N; ++k) f[k] = fminf(fmaxf(f[k] / last, 0.0f), 1.0f);
    f[N - 1] = 1.0f;
  }
  __syncthreads();
}

__global__ __launch_bounds__(NT) void strictly_increasing_kernel(const float* __restrict__ F, int N, double eps,
                                                                 float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  const int b = blockIdx.x;
  for (int k = threadIdx.x; k < N; k += blockDim.x) smem_f[k] = F[(size_t)b * N + k];
  __syncthreads();
  strictly_increasing_block(smem_f, N, eps);
  for (int k = threadIdx.x; k < N; k += blockDim.x) out[(size_t)b * N + k] = smem_f[k];
}

// resample_cdf (MN/checkpoint_utils.py:53-62): repair -> F.interpolate(linear, align_corners=True) -> repair
__global__ __launch_bounds__(NT) void resample_cdf_kernel(const float* __restrict__ F, int N, int L,
                                                          float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  float* a = smem_f;        // N
  float* u = smem_f + N;    // L
  const int b = blockIdx.x;
  for (int k = threadIdx.x; k < N; k += blockDim.x) a[k] = F[(size_t)b * N + k];
  __syncthreads();
  strictly_increasing_block(a, N, 1e-4);
  // ATen upsample_linear1d, align_corners=True: scale = (N-1)/(L-1) in float32
  const float scale = (L > 1) ? (float)(N - 1) / (float)(L - 1) : 0.0f;
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    const float src = fmul(scale, (float)i);
    const int i0 = min((int)src, N - 1);
    const int i1 = min(i0 + 1, N - 1);
    const float l1 = fsub(src, (float)i0), l0 = fsub(1.0f, l1);
    u[i] = fadd(fmul(l0, a[i0]), fmul(l1, a[i1]));
  }
  __syncthreads();
  strictly_increasing_block(u, L, 1e-4);
  for (int k = threadIdx.x; k < L; k += blockDim.x) out[(size_t)b * L + k] = u[k];
}

}  // namespace attwarp

using namespace attwarp;

extern "C" int attwarp_axis_map_from_cdf(const float* F, int B, int L, int n_out, float* map, void* stream) {
  ATTWARP_REQUIRE(F && map, "axis_map_from_cdf: null pointer");
  ATTWARP_REQUIRE(B > 0 && L > 0 && n_out > 0, "axis_map_from_cdf: non-positive size");
  if (L > MAX_L) return fail(ATTWARP_E_UNSUPPORTED, "axis_map_from_cdf: L=%d > %d", L, MAX_L);
  const size_t lds = (size_t)(L + 1) * sizeof(double);
  hipLaunchKernelGGL(axis_map_from_cdf_kernel, dim3(B), dim3(NT), lds, as_stream(stream), F, L, n_out, map);
  return check_launch("axis_map_from_cdf_kernel");
}

extern "C" int attwarp_cdf_from_density(const float* p, int B, int L, float* F, void* stream) {
  ATTWARP_REQUIRE(p && F, "cdf_from_density: null pointer");
  ATTWARP_REQUIRE(B > 0 && L > 0, "cdf_from_density: non-positive size");
  if (L > MAX_L) return fail(ATTWARP_E_UNSUPPORTED, "cdf_from_density: L=%d > %d", L, MAX_L);
  const size_t lds = 8 * sizeof(double) + (size_t)L * sizeof(float);
  hipLaunchKernelGGL(cdf_from_density_kernel, dim3(B), dim3(NT), lds, as_stream(stream), p, L, F);
  return check_launch("cdf_from_density_kernel");
}

extern "C" int attwarp_upsample_pdf_right_inverse(const float* y, int N, int Lo, int L, const double* inv, float* out,
                                                  void* stream) {
  ATTWARP_REQUIRE(y && inv && out, "upsample_pdf_right_inverse: null pointer");
  ATTWARP_REQUIRE(N > 0 && Lo > 0 && L > 0, "upsample_pdf_right_inverse: non-positive size");
  if (Lo > 64) return fail(ATTWARP_E_UNSUPPORTED, "upsample_pdf_right_inverse: L_out=%d > 64", Lo);
  hipLaunchKernelGGL(right_inverse_kernel, dim3(N), dim3(NT), 0, as_stream(stream), y, Lo, L, inv, out);
  return check_launch("right_inverse_kernel");
}

extern "C" int attwarp_axis_maps_from_pdf(const float* px, const float* py, int B, int Lo, int W, int H, int W_out,
                                          int H_out, const double* inv_x, const double* inv_y, float* map_x,
                                          float* map_y, void* stream) {
  ATTWARP_REQUIRE(px && py && inv_x && inv_y && map_x && map_y, "axis_maps_from_pdf: null pointer");
  ATTWARP_REQUIRE(B > 0 && Lo > 0 && W > 0 && H > 0 && W_out > 0 && H_out > 0, "axis_maps_from_pdf: non-positive size");
  if (Lo > 64) return fail(ATTWARP_E_UNSUPPORTED, "axis_maps_from_pdf: L_out=%d > 64", Lo);
  const int L = W > H ? W : H;
  if (L > 8192) return fail(ATTWARP_E_UNSUPPORTED, "axis_maps_from_pdf: max(W,H)=%d > 8192", L);
  const size_t lds = (size_t)(8 + L + 2) * sizeof(double) + (size_t)L * sizeof(float);
  if (const int rc = grant_dynamic_lds(axis_maps_from_pdf_kernel, lds, "axis_maps_from_pdf")) return rc;
  hipLaunchKernelGGL(axis_maps_from_pdf_kernel, dim3(B, 2), dim3(NT), lds, as_stream(stream), px, py, Lo, W, H, W_out,
                     H_out, inv_x, inv_y, map_x, map_y);
  return check_launch("axis_maps_from_pdf_kernel");
}

extern "C" int attwarp_axis_maps_from_steps_t(const void* steps, int dtype, int T, int B, int g, int W, int H, int W_out,
                                              int H_out, const double* inv_x, const double* inv_y, float* map_x,
                                              float* map_y, float* att_out, void* stream) {
  ATTWARP_REQUIRE(steps && inv_x && inv_y && map_x && map_y, "axis_maps_from_steps: null pointer");
  ATTWARP_REQUIRE(T > 0 && B > 0 && g > 0 && W > 0 && H > 0 && W_out > 0 && H_out > 0,
                  "axis_maps_from_steps: non-positive size");
  ATTWARP_REQUIRE(dtype == ATTWARP_F32 || dtype == ATTWARP_F16 || dtype == ATTWARP_BF16,
                  "axis_maps_from_steps: dtype must be F32, F16 or BF16 (got %d)", dtype);
  if (g > 64) return fail(ATTWARP_E_UNSUPPORTED, "axis_maps_from_steps: grid side %d > 64", g);
  const int L = W > H ? W : H;
  if (L > 8192) return fail(ATTWARP_E_UNSUPPORTED, "axis_maps_from_steps: max(W,H)=%d > 8192", L);
  if (B > 65535) return fail(ATTWARP_E_UNSUPPORTED, "axis_maps_from_steps: B > 65535");
  StepsMapsArgs a;
  a.steps = steps; a.step_dtype = dtype; a.T = T; a.B = B; a.g = g; a.W = W; a.H = H; a.W_out = W_out; a.H_out = H_out;
  a.inv_x = inv_x; a.inv_y = inv_y; a.map_x = map_x; a.map_y = map_y; a.att_out = att_out;
  const dim3 grid(B, 2), blk(NT);
  const size_t lds = steps_maps_lds_bytes(L, g);
  if (const int rc = dtype == ATTWARP_F32 ? grant_dynamic_lds(axis_maps_from_steps_kernel<float>, lds, "axis_maps_from_steps")
                     : dtype == ATTWARP_F16 ? grant_dynamic_lds(axis_maps_from_steps_kernel<__half>, lds, "axis_maps_from_steps")
                                            : grant_dynamic_lds(axis_maps_from_steps_kernel<__hip_bfloat16>, lds, "axis_maps_from_steps"))
    return rc;
  if (dtype == ATTWARP_F32) hipLaunchKernelGGL(axis_maps_from_steps_kernel<float>, grid, blk, lds, as_stream(stream), a);
  else if (dtype == ATTWARP_F16) hipLaunchKernelGGL(axis_maps_from_steps_kernel<__half>, grid, blk, lds, as_stream(stream), a);
  else hipLaunchKernelGGL(axis_maps_from_steps_kernel<__hip_bfloat16>, grid, blk, lds, as_stream(stream), a);
  return check_launch("axis_maps_from_steps_kernel");
}

extern "C" int attwarp_axis_maps_from_steps(const float* steps, int T, int B, int g, int W, int H, int W_out, int H_out,
                                            const double* inv_x, const double* inv_y, float* map_x, float* map_y,
                                            float* att_out, void* stream) {
  return attwarp_axis_maps_from_steps_t(steps, ATTWARP_F32, T, B, g, W, H, W_out, H_out, inv_x, inv_y, map_x, map_y, att_out,
                                        stream);
}

extern "C" int attwarp_safe_softmax(const float* logits, int B, int N, float eps, float* out, void* stream) {
  ATTWARP_REQUIRE(logits && out, "safe_softmax: null pointer");
  ATTWARP_REQUIRE(B > 0 && N > 0, "safe_softmax: non-positive size");
  if (N > MAX_L) return fail(ATTWARP_E_UNSUPPORTED, "safe_softmax: N=%d > %d", N, MAX_L);
  const size_t lds = 8 * sizeof(double) + (size_t)N * sizeof(float);
  hipLaunchKernelGGL(safe_softmax_kernel, dim3(B), dim3(NT), lds, as_stream(stream), logits, N, eps, out);
  return check_launch("safe_softmax_kernel");
}

extern "C" int attwarp_make_strictly_increasing(const float* F, int B, int N, double eps, float* out, void* stream) {
  ATTWARP_REQUIRE(F && out, "make_strictly_increasing: null pointer");
  ATTWARP_REQUIRE(B > 0 && N > 0, "make_strictly_increasing: non-positive size");
  if (N > MAX_L) return fail(ATTWARP_E_UNSUPPORTED, "make_strictly_increasing: N=%d > %d", N, MAX_L);
  hipLaunchKernelGGL(strictly_increasing_kernel, dim3(B), dim3(NT), (size_t)N * sizeof(float), as_stream(stream), F, N,
                     eps, out);
  return check_launch("strictly_increasing_kernel");
}

extern "C" int attwarp_resample_cdf(const float* F, int B, int N, int L, float* out, void* stream) {
  ATTWARP_REQUIRE(F && out, "resample_cdf: null pointer");
  ATTWARP_REQUIRE(B > 0 && N > 0 && L > 0, "resample_cdf: non-positive size");
  if (N + L > 2 * MAX_L) return fail(ATTWARP_E_UNSUPPORTED, "resample_cdf: N+L=%d too large", N + L);
  hipLaunchKernelGGL(resample_cdf_kernel, dim3(B), dim3(NT), (size_t)(N + L) * sizeof(float), as_stream(stream), F, N,
                     L, out);
  return check_launch("resample_cdf_kernel");
}
