// Per-axis 1-D device blocks shared by axis.hip and the fused step kernel (remap_rows_kernel.hpp): A8 right-inverse PDF
// up-sample, A9 PDF -> CDF, A11 inverse map (np.interp), and the A2+A6+A8+A9+A11 chain from per-step attention maps.
// Every function is called by ALL threads of a 256-thread workgroup; scans are sequential where the reference's are
// (see axis.hip).
#pragma once
#include "common.hpp"
#include "interp.hpp"

namespace attwarp {

constexpr int AXIS_NT = 256;

// ---- A11 core: CDF (float, LDS or global) -> knots -> map ------------------------------
// MN/checkpoint_utils.py:167-193.  xn: LDS double[L+1].
__device__ void map_from_cdf_block(const float* F, int L, int n_out, double* xn, float* map) {
  const int len = L + 1;
  for (int k = threadIdx.x; k < len; k += blockDim.x)
    xn[k] = (k == 0) ? 0.0 : (double)F[k - 1] * (double)n_out;   // concatenate(([0.0], F)) * float(n_out)
  __syncthreads();
  if (threadIdx.x == 0) xn[len - 1] = (double)n_out;              // x_new_map_fwd[-1] = W_out
  __syncthreads();
  int tie = 0;
  for (int k = threadIdx.x; k + 1 < len; k += blockDim.x) tie |= ((xn[k + 1] - xn[k]) <= 0.0);
  tie = __syncthreads_or(tie);
  if (tie) {
    // += (1e-4 / max(W_out,1)) * np.arange(size, dtype=float32): python scalar * float32 array
    // is a float32 product (then promoted to float64 by the in-place add)
    const float c = (float)(1e-4 / (double)max(n_out, 1));
    for (int k = threadIdx.x; k < len; k += blockDim.x) xn[k] += (double)fmul(c, (float)k);
    __syncthreads();
  }
  const bool mono = block_is_sorted(xn, len);
  np_interp_block(xn, len, n_out, map, mono);
}

// ---- A9 core: density (LDS float, in place) -> CDF -----------------------------------
// MN/checkpoint_utils.py:30-41.  p: LDS float[L], overwritten by the CDF.  red: LDS double[NT/64].
__device__ void cdf_from_density_block(float* p, int L, double* red) {
  double acc = 0.0;
  for (int k = threadIdx.x; k < L; k += blockDim.x) {
    float v = p[k];
    v = (isnan(v) || isinf(v)) ? 0.0f : fmaxf(v, 0.0f);   // clamp_min(0) then nan_to_num(->0)
    p[k] = v;
    acc += (double)v;
  }
  const float denom = fmaxf((float)block_sum(acc, red), 1e-6f);
  __syncthreads();
  for (int k = threadIdx.x; k < L; k += blockDim.x) p[k] = p[k] / denom;
  __syncthreads();
  // Running sum in double (what torch's CPU cumsum does), rounded to float32 per prefix.
  // The values are float32 in [0,1] summing to ~1.  If every non-zero value is >= 2^-28, every partial sum
  // is a multiple of 2^-51 below 2 and therefore EXACT in double: any association gives the same bits, so
  // a parallel scan equals the sequential one.  Otherwise (denormal-ish densities) one lane runs the
  // sequential scan.
  int small = 0;
  double lsum = 0.0;
  const int per = (L + (int)blockDim.x - 1) / (int)blockDim.x;      // consecutive elements per thread
  const int k0 = threadIdx.x * per, k1 = min(k0 + per, L);
  for (int k = k0; k < k1; ++k) {
    const float v = p[k];
    small |= (v != 0.0f) && (v < 3.7252902984619140625e-9f);          // 2^-28
    lsum += (double)v;
  }
  small = __syncthreads_or(small || !(lsum < 2.0));
  if (!small) {
    // exclusive scan of the per-thread sums: wave shuffle scan, then the wave totals through LDS
    double inc = lsum;
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
      const double t = __shfl_up(inc, o, WAVE);
      if (lane >= o) inc += t;
    }
    if (lane == WAVE - 1) red[wid] = inc;
    __syncthreads();
    double base = inc - lsum;
    for (int w = 0; w < wid; ++w) base += red[w];
    double c = base;
    for (int k = k0; k < k1; ++k) {
      c += (double)p[k];
      p[k] = (float)c;
    }
    __syncthreads();
    if (threadIdx.x == 0) p[L - 1] = 1.0f;
    __syncthreads();
    return;
  }
  if (threadIdx.x == 0) {
    double c = 0.0;
    for (int k = 0; k < L; ++k) {
      c += (double)p[k];
      p[k] = (float)c;
    }
    p[L - 1] = 1.0f;
  }
  __syncthreads();
}

// ---- A8 core: right-inverse up-sample ---------------------------------------------------
// MN/checkpoint_utils.py:64-131.  y: Lo floats (global), inv: Lo x Lo doubles (global),
// tmp: LDS float[Lo], out: float[L] (LDS or global).
__device__ __forceinline__ void adaptive_window(int k, int L, int Lo, int& s, int& e) {
  // k < Lo <= 64 and L <= 16384: the products fit 32 bits
  s = (int)(((unsigned)k * (unsigned)L) / (unsigned)Lo);
  e = (int)(((unsigned)(k + 1) * (unsigned)L + (unsigned)Lo - 1u) / (unsigned)Lo);
}

__device__ void right_inverse_block(const float* y, int Lo, int L, const double* inv, float* tmp, float* out,
                                    bool clamp0) {
  for (int k = threadIdx.x; k < Lo; k += blockDim.x) {
    double acc = 0.0;
    const double* row = inv + (size_t)k * Lo;
    int j = 0;
    for (; j + 8 <= Lo; j += 8) {          // 8 products in flight (loads + multiplies), then the ordered sum
      double pr[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) pr[u] = (double)y[j + u] * row[j + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = acc + pr[u];
    }
    for (; j < Lo; ++j) acc = acc + (double)y[j] * row[j];
    tmp[k] = (float)acc;
  }
  __syncthreads();
  for (int l = threadIdx.x; l < L; l += blockDim.x) {
    const int k0 = (int)(((unsigned)l * (unsigned)Lo) / (unsigned)L);
    float x = 0.0f;
    for (int k = max(k0 - 1, 0); k <= min(k0 + 1, Lo - 1); ++k) {
      int s, e;
      adaptive_window(k, L, Lo, s, e);
      if (l >= s && l < e) {
        const float a = 1.0f / (float)max(e - s, 1);
        x = fadd(x, fmul(tmp[k], a));
      }
    }
    out[l] = clamp0 ? fmaxf(x, 0.0f) : x;
  }
  __syncthreads();
}

// ---- A2 + A6 + A8 + A9 + A11 fused: per-step attention maps -> inverse maps, one launch -------------
// steps [T,B,g*g] float32 (A1 output) -> mean over steps (A2, llava.py:409-411) -> marginals of the
// g x g map (A6) -> right-inverse up-sample, clamp, CDF, inverse map (as axis_maps_from_pdf_kernel).
// Bit-identical to running the stages one by one.  grid = (B, 2).
struct StepsMapsArgs {
  const void* steps;       // [T,B,g*g] in the attention dtype (float32 / float16 / bfloat16: what A1 wrote)
  int step_dtype;          // ATTWARP_F32 / F16 / BF16
  int T, B, g, W, H, W_out, H_out;
  const double* inv_x;     // [g,g] cached inverse of A A^T + eps I for L = W
  const double* inv_y;     // ... for L = H
  float* map_x;            // [B,W_out]
  float* map_y;            // [B,H_out]
  float* att_out;          // optional [B,g*g]
};
// LDS the block needs (bytes) for axis length L and a g x g grid
inline size_t steps_maps_lds_bytes(int L, int g) {
  return (size_t)(8 + L + 2) * sizeof(double) + (size_t)(L + g * g + 2) * sizeof(float) + (size_t)g * g * sizeof(double);
}
// One 256-thread workgroup per (sample b, axis).  smem_d: steps_maps_lds_bytes(); tmp, pm: 64 floats of LDS each.
// TC: steps of a token requested in one go (24: one memory round trip for T <= 24; the fused step kernel uses 8 to keep
// its register allocation at the resample's)
// ST: dtype of the step maps.  A2 (llava.py:409-411) runs in the model dtype: float64 accumulation in step order, ONE
// rounding to ST, the division by T in ST (attn_finalize_kernel's arithmetic); everything after it is float32 as in
// gt_marginals of the float() of that map.
template <int TC, typename ST>
__device__ __forceinline__ void axis_maps_from_steps_block(const StepsMapsArgs& a, int b, int axis, double* smem_d,
                                                           float* tmp, float* pm) {
  const ST* __restrict__ steps = static_cast<const ST*>(a.steps);
  const int T = a.T, B = a.B, g = a.g;
  float* __restrict__ att_out = a.att_out;
  const int L = axis ? a.H : a.W, n_out = axis ? a.H_out : a.W_out, ntok = g * g;
  const double* inv = axis ? a.inv_y : a.inv_x;
  float* map = (axis ? a.map_y : a.map_x) + (size_t)b * n_out;
  double* red = smem_d;                         // 8
  double* xn = smem_d + 8;                      // L+1
  float* p = reinterpret_cast<float*>(xn + L + 1 + ((L + 1) & 1));   // L floats
  float* att = p + L;                           // g*g floats
  double* s_inv = reinterpret_cast<double*>(p + ((L + ntok + 1) & ~1));   // g*g doubles: this axis' inverse matrix
  // the inverse matrix is needed only after the marginals: request it first, it arrives behind the step maps
  constexpr int IPT = 4;                        // g <= 32: g*g <= 1024 = 4 per thread (larger grids: strided loop)
  constexpr int NT = AXIS_NT;
  double ireg[IPT];
  const bool inv_regs = ntok <= IPT * NT;
  if (inv_regs) {
#pragma unroll
    for (int u = 0; u < IPT; ++u) {
      const int i = threadIdx.x + NT * u;
      ireg[u] = i < ntok ? inv[i] : 0.0;
    }
  }
  // A2: mean over generation steps (float64 accumulate in step order, one rounding, float32 divide).  A thread owns
  // tokens tid, tid + 256, tid + 512; ALL their loads of a chunk of 24 steps are in flight at once (one memory round
  // trip for T <= 24 instead of 3 tokens x 3 batches), then the ordered accumulation.
  {
    constexpr int TPT = 3;
    const size_t tstride = (size_t)B * ntok;
    for (int i0 = 0; i0 < ntok; i0 += TPT * NT) {
      double acc[TPT];
#pragma unroll
      for (int u = 0; u < TPT; ++u) acc[u] = 0.0;
      for (int t0 = 0; t0 < T; t0 += TC) {
        ST v[TPT][TC];
#pragma unroll
        for (int u = 0; u < TPT; ++u) {
          const int i = i0 + threadIdx.x + NT * u;
          const ST* sp = steps + (size_t)b * ntok + min(i, ntok - 1);
#pragma unroll
          for (int j = 0; j < TC; ++j) v[u][j] = sp[(size_t)min(t0 + j, T - 1) * tstride];
        }
#pragma unroll
        for (int u = 0; u < TPT; ++u)
#pragma unroll
          for (int j = 0; j < TC; ++j)
            if (t0 + j < T) acc[u] += (double)to_f32<ST>(v[u][j]);
      }
#pragma unroll
      for (int u = 0; u < TPT; ++u) {
        const int i = i0 + threadIdx.x + NT * u;
        if (i < ntok) {
          const float m = to_f32<ST>(div_t<ST>(from_f64<ST>(acc[u]), from_f32<ST>((float)T)));
          att[i] = m;
          if (att_out && axis == 0) att_out[(size_t)b * ntok + i] = m;
        }
      }
    }
  }
  if (inv_regs) {
#pragma unroll
    for (int u = 0; u < IPT; ++u) {
      const int i = threadIdx.x + NT * u;
      if (i < ntok) s_inv[i] = ireg[u];
    }
  } else {
    for (int i = threadIdx.x; i < ntok; i += NT) s_inv[i] = inv[i];
  }
  __syncthreads();
  // A6: marginal along this axis (x: sum over rows; y: sum over columns), clamp >= 0, normalise
  for (int k = threadIdx.x; k < g; k += blockDim.x) {
    double acc = 0.0;
    int j = 0;
    for (; j + 8 <= g; j += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = axis ? att[k * g + j + u] : att[(j + u) * g + k];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += (double)((v[u] != v[u]) ? v[u] : (v[u] > 0.0f ? v[u] : 0.0f));
    }
    for (; j < g; ++j) {
      const float v = axis ? att[k * g + j] : att[j * g + k];
      acc += (double)((v != v) ? v : (v > 0.0f ? v : 0.0f));
    }
    pm[k] = (float)acc;
  }
  __syncthreads();
  {   // every thread computes the same total (no extra barrier, LDS broadcast reads)
    double tot = 0.0;
    int k = 0;
    for (; k + 8 <= g; k += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = pm[k + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) tot += (double)v[u];
    }
    for (; k < g; ++k) tot += (double)pm[k];
    const float totf = fmaxf((float)tot, 1e-6f);
    __syncthreads();
    if (threadIdx.x < g) pm[threadIdx.x] = pm[threadIdx.x] / totf;
  }
  __syncthreads();
  right_inverse_block(pm, g, L, s_inv, tmp, p, true);
  cdf_from_density_block(p, L, red);
  map_from_cdf_block(p, L, n_out, xn, map);
}


}  // namespace attwarp
