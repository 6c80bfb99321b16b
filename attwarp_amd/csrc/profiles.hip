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
#include "profiles_blocks.hpp"

namespace attwarp {

constexpr int NT = PROF_NT;

constexpr int RBAND = 32;     // rows per LDS tile
constexpr int TSTR = 128 + 8; // tile row stride in doubles: 8 rows x 8 accumulators spread evenly over the banks

// ---- single pass over [B,H,W]: column sums AND per-leaf row sums ---------------------------------
// grid = (nleaves, B); block = 256.  The block owns one leaf (a strip of <= 128 columns) and walks all H rows
// in bands of 32 through an LDS tile of TRANSFORMED values (double, row major): the element transform (clamp,
// square / sqrt / exp / log, bias) runs once per element, by the thread that loaded it.
//   load     : thread (quad q = tid & 31, row tid >> 5 (+8 per pass)) reads 4 consecutive columns with one
//              vector load when the rows are 4-element aligned; the NEXT band's loads are issued before this
//              band is reduced;
//   columns  : thread c < len keeps the running sum of column c over rows in ascending order = the order
//              np.sum(axis=0) uses, so the x profile is bit-identical to numpy's;
//   rows     : 8 lanes per row own numpy's 8 strided accumulators of the leaf, an xor butterfly over them is
//              exactly ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), lane 0 adds the tail.
// leaf sums go to ls[b][row][leaf]; the per-row tree over leaves runs in the finalize kernels.
template <typename T>
struct alignas(4 * sizeof(T)) Quad4 {
  T v[4];
};

template <typename T, typename XF>
__global__ __launch_bounds__(NT) void profiles_kernel(const T* __restrict__ A, int H, int W, XF xf,
                                                      const PairwisePlan P, double* __restrict__ col,
                                                      double* __restrict__ ls, int vec_ok) {
  __shared__ __attribute__((aligned(16))) double tile[RBAND * TSTR];
  const int leaf = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int off = P.off[leaf], len = P.len[leaf], nleaves = P.nleaves;
  const T* base = A + (size_t)b * H * W + off;
  const int qc = 4 * (tid & 31), qr = tid >> 5;          // this thread's 4 columns / first row of a band
  const int rrow = tid >> 3, rk = tid & 7;               // row task: (row of the band, accumulator)
  const int len8 = len - (len % 8);
  constexpr int NPASS = RBAND / 8;
  Quad4<T> raw[NPASS];

  // (vector loads may run up to 3 elements past the strip inside the row: W % 4 == 0 keeps them inside the image)
#define ATTWARP_PROFILES_FETCH(row0_)                                                        \
  _Pragma("unroll") for (int ps = 0; ps < NPASS; ++ps) {                                      \
    const int r_ = (row0_) + qr + 8 * ps;                                                    \
    if (r_ < H && qc < len) {                                                                \
      const T* src_ = base + (size_t)r_ * W + qc;                                            \
      if (vec_ok) {                                                                          \
        raw[ps] = *reinterpret_cast<const Quad4<T>*>(src_);                                  \
      } else {                                                                               \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) raw[ps].v[j] = (qc + j < len) ? src_[j] : T(0); \
      }                                                                                      \
    }                                                                                        \
  }
  ATTWARP_PROFILES_FETCH(0)
  double cacc = 0.0;
  for (int row0 = 0; row0 < H; row0 += RBAND) {
    const int nb = min(RBAND, H - row0);
    if (qc < len) {
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const int r = qr + 8 * ps;
        if (r < nb) {
          double* d = tile + r * TSTR + qc;
#pragma unroll
          for (int j = 0; j < 4; ++j) d[j] = xf((double)raw[ps].v[j]);
        }
      }
    }
    __syncthreads();
    if (row0 + RBAND < H) ATTWARP_PROFILES_FETCH(row0 + RBAND)
    // rows: numpy's leaf sum of row rrow over the strip
    {
      const double* tr = tile + rrow * TSTR;
      double r = 0.0;
      if (rrow < nb && len >= 8) {
        r = tr[rk];
        for (int j = 8; j < len8; j += 8) r += tr[j + rk];
      }
      r += __shfl_xor(r, 1, WAVE);
      r += __shfl_xor(r, 2, WAVE);
      r += __shfl_xor(r, 4, WAVE);
      if (rrow < nb && rk == 0) {
        if (len < 8) {
          r = 0.0;
          for (int j = 0; j < len; ++j) r += tr[j];
        } else {
          for (int j = len8; j < len; ++j) r += tr[j];
        }
        ls[((size_t)b * H + row0 + rrow) * nleaves + leaf] = r;
      }
    }
    // columns: ascending rows
    if (tid < len) {
      const double* tc = tile + tid;
      for (int r = 0; r < nb; ++r) cacc = cacc + tc[r * TSTR];
    }
    __syncthreads();
  }
#undef ATTWARP_PROFILES_FETCH
  if (tid < len) col[(size_t)b * W + off + tid] = cacc;
}

// ---- uint8 attention (the main_batched chain): body profiles_u8_block (profiles_blocks.hpp).  grid = (leaf, image) ----
template <int TR, bool UA>
__global__ __launch_bounds__(NT) void profiles_u8_kernel(const uint8_t* __restrict__ A, int H, int W,
                                                         XfAttention<TR> xf, const PairwisePlan P,
                                                         double* __restrict__ col, double* __restrict__ ls) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[profiles_u8_lds_bytes<TR>()];
  const int leaf = blockIdx.x, b = blockIdx.y;
  profiles_u8_block<TR, UA>(A + (size_t)b * H * W, H, W, xf, P.off[leaf], P.len[leaf], P.nleaves, leaf, col + (size_t)b * W,
                            ls + (size_t)b * H * P.nleaves, lds);
}
template <int TR>
static int launch_profiles_u8(const void* A, int B, int H, int W, XfAttention<TR> xf, const PairwisePlan& P,
                              double* col, double* ls, hipStream_t st, bool* handled) {
  *handled = false;
  if (tune(TUNE_PROFILES_VARIANT) == 1) return ATTWARP_OK;
  for (int j = 0; j < P.nleaves; ++j)
    if (P.len[j] < 8) return ATTWARP_OK;
  *handled = true;
  // rows that do not start on dword boundaries (W % 4 != 0, or a view that starts anywhere): the UA form of the block
  const bool ua = W % 4 != 0 || (reinterpret_cast<uintptr_t>(A) & 3u) != 0;
  if (ua) hipLaunchKernelGGL((profiles_u8_kernel<TR, true>), dim3(P.nleaves, B), dim3(NT), 0, st, (const uint8_t*)A, H, W, xf, P, col, ls);
  else hipLaunchKernelGGL((profiles_u8_kernel<TR, false>), dim3(P.nleaves, B), dim3(NT), 0, st, (const uint8_t*)A, H, W, xf, P, col, ls);
  return check_launch("profiles_u8_kernel");
}

// ---- float32 input with a cheap transform (gt_marginals' clamp; identity / square attention) --------------------
// profiles_kernel<float, XF> keeps a band of TRANSFORMED doubles in LDS (8 bytes per element written once and read
// twice) and reached 3.6 TB/s = 0.45 of the HBM peak on B=256 x 1024 x 1024 (profiles/round2_stage_bench.txt).  Same
// outputs and summation orders here with the profiles_u8_kernel structure: LDS holds the RAW floats of a band of 64
// rows x the leaf's <= 128 columns (one buffer, the next band waits in registers: 128 bytes per lane in flight), the
// transform runs in registers where a value is consumed, waves 0-1 own the row sums ((row, half): 16 x ds_read_b128,
// four float64 chains, one xor-1 shuffle), waves 2-3 the column sums (ascending rows).  Row stride 136 dwords: the 16
// lanes of a ds_read_b128 phase (8 rows x 2 halves) cover all 64 banks.  Requires W % 4 == 0, a 16-byte aligned
// base and every leaf >= 8 long and a multiple of 4.
constexpr int F32_RB = 64;
constexpr int F32_STR = 128 + 8;       // dwords

template <typename XF>
__global__ __launch_bounds__(NT) void profiles_f32_kernel(const float* __restrict__ A, int H, int W, XF xf,
                                                          const PairwisePlan P, double* __restrict__ col,
                                                          double* __restrict__ ls) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) float tile[F32_RB * F32_STR];
  const int tid = threadIdx.x, b = blockIdx.y, leaf = blockIdx.x;
  const int coff = P.off[leaf], len = P.len[leaf], nleaves = P.nleaves;
  const int m = len >> 3;                                          // steps of the stride-8 accumulators
  const float* base = A + (size_t)b * H * W + coff;
  const int gq = tid & 31, gr = tid >> 5;                          // this thread's quad / first row of a band
  const int nq = len >> 2;
  constexpr int NPASS = F32_RB / 8;
  v4f raw[NPASS];
#define ATTWARP_F32P_FETCH(row0_)                                                                \
  _Pragma("unroll") for (int ps = 0; ps < NPASS; ++ps) {                                          \
    const int r_ = (row0_) + gr + 8 * ps;                                                        \
    if (r_ < H && gq < nq) raw[ps] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(base + (size_t)r_ * W) + gq); \
  }
  ATTWARP_F32P_FETCH(0)
  double cacc = 0.0;
  for (int row0 = 0; row0 < H; row0 += F32_RB) {
    const int nb = min(F32_RB, H - row0);
    if (gq < nq) {
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) reinterpret_cast<v4f*>(tile + (gr + 8 * ps) * F32_STR)[gq] = raw[ps];
    }
    __syncthreads();
    if (row0 + F32_RB < H) ATTWARP_F32P_FETCH(row0 + F32_RB)
    if (tid < 2 * F32_RB) {
      // ---- rows (waves 0-1): thread = (row, half) ----
      const int r = tid >> 1, hh = tid & 1;
      const float* rp = tile + r * F32_STR + 4 * hh;
      v4f wv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) wv[i] = *reinterpret_cast<const v4f*>(rp + 8 * i);   // in-row reads past the leaf: unused
      double a0 = xf.from_f32(wv[0].x), a1 = xf.from_f32(wv[0].y), a2 = xf.from_f32(wv[0].z), a3 = xf.from_f32(wv[0].w);
#pragma unroll
      for (int i = 1; i < 16; ++i) {
        if (i < m) {                                               // block uniform
          a0 += xf.from_f32(wv[i].x); a1 += xf.from_f32(wv[i].y); a2 += xf.from_f32(wv[i].z); a3 += xf.from_f32(wv[i].w);
        }
      }
      double u = (a0 + a1) + (a2 + a3);
      u = u + __shfl_xor(u, 1, WAVE);                              // ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7))
      if (hh == 0 && r < nb) {
        const float* tp = tile + r * F32_STR;
        for (int i = 8 * m; i < len; ++i) u += xf.from_f32(tp[i]);
        ls[((size_t)b * H + row0 + r) * nleaves + leaf] = u;
      }
    } else if (tid - 2 * F32_RB < len) {
      // ---- columns (waves 2-3): ascending rows ----
      const float* cp = tile + (tid - 2 * F32_RB);
      for (int r0 = 0; r0 < nb; r0 += 16) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = cp[(r0 + i) * F32_STR];   // rows >= nb of a partial band: stale, unused
        if (r0 + 16 <= nb) {
#pragma unroll
          for (int i = 0; i < 16; ++i) cacc = cacc + xf.from_f32(v[i]);
        } else {
          for (int i = 0; i < nb - r0; ++i) cacc = cacc + xf.from_f32(v[i]);
        }
      }
    }
    __syncthreads();                                               // the tile is rewritten by the next band
  }
#undef ATTWARP_F32P_FETCH
  if (tid >= 2 * F32_RB && tid - 2 * F32_RB < len) col[(size_t)b * W + coff + tid - 2 * F32_RB] = cacc;
}

template <typename XF>
static int launch_profiles_f32(const void* A, int B, int H, int W, XF xf, const PairwisePlan& P, double* col, double* ls,
                               hipStream_t st, bool* handled) {
  *handled = false;
  if (tune(TUNE_PROFILES_VARIANT) == 1 || W % 4 != 0 || (reinterpret_cast<uintptr_t>(A) & 15u) != 0) return ATTWARP_OK;
  for (int j = 0; j < P.nleaves; ++j)
    if (P.len[j] < 8 || P.len[j] % 4 != 0) return ATTWARP_OK;
  *handled = true;
  hipLaunchKernelGGL((profiles_f32_kernel<XF>), dim3(P.nleaves, B), dim3(NT), 0, st, (const float*)A, H, W, xf, P, col, ls);
  return check_launch("profiles_f32_kernel");
}

// ---- A6 finalize: normalise a marginal.  grid = (B, 2) ----------------------------------
__global__ __launch_bounds__(NT) void marginals_finalize_kernel(const double* __restrict__ col,
                                                                const double* __restrict__ ls, int nleaves, int H,
                                                                int W, float* __restrict__ px,
                                                                float* __restrict__ py) {
  extern __shared__ __attribute__((aligned(16))) double smem_d[];   // n doubles
  __shared__ double red[NT / WAVE];
  const int b = blockIdx.x, axis = blockIdx.y;
  const int n = axis ? H : W;
  float* dst = (axis ? py : px) + (size_t)b * n;
  double acc = 0.0;
  for (int k = threadIdx.x; k < n; k += blockDim.x) {
    double v;
    if (axis) {
      v = 0.0;
      const double* l = ls + ((size_t)b * H + k) * nleaves;
      for (int j = 0; j < nleaves; ++j) v += l[j];
    } else {
      v = col[(size_t)b * W + k];
    }
    smem_d[k] = v;
    acc += (double)(float)v;
  }
  const float tsum = (float)block_sum(acc, red);
  const float tot = (tsum != tsum) ? tsum : fmaxf(tsum, 1e-6f);       // torch's clamp_min propagates NaN (fmaxf drops it)
  for (int k = threadIdx.x; k < n; k += blockDim.x) dst[k] = (float)smem_d[k] / tot;
}

// ---- A5: adaptive average pool.  grid = (oh, B), one band of rows per block ---------------
__global__ __launch_bounds__(NT) void adaptive_pool_kernel(const float* __restrict__ A, int H, int W, int oh, int ow,
                                                           int sanitize, int vec_ok, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) double colsum[];   // W doubles
  const int b = blockIdx.y, i = blockIdx.x;
  const int ys = (int)(((long long)i * H) / oh);
  const int ye = (int)((((long long)(i + 1)) * H + oh - 1) / oh);
  const float* base = A + (size_t)b * H * W;
  if (vec_ok) {
    // 16 bytes per lane and row, 8 rows in flight: the dword form of this loop below reached 0.61 of the HBM peak on
    // B=256 x 1024 x 1024 (profiles/round2_stage_bench.txt); same summation order (rows ascending per column)
    for (int q = threadIdx.x; 4 * q < W; q += blockDim.x) {
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      typedef float v4f __attribute__((ext_vector_type(4)));
      const v4f* src = reinterpret_cast<const v4f*>(base) + q;
      const size_t rs = (size_t)W / 4;
      int r = ys;
      for (; r + 8 <= ye; r += 8) {
        v4f v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(src + (size_t)(r + k) * rs);
#pragma unroll
        for (int k = 0; k < 8; ++k) { a0 += (double)v[k].x; a1 += (double)v[k].y; a2 += (double)v[k].z; a3 += (double)v[k].w; }
      }
      for (; r < ye; ++r) {
        const v4f v = __builtin_nontemporal_load(src + (size_t)r * rs);
        a0 += (double)v.x; a1 += (double)v.y; a2 += (double)v.z; a3 += (double)v.w;
      }
      colsum[4 * q] = a0; colsum[4 * q + 1] = a1; colsum[4 * q + 2] = a2; colsum[4 * q + 3] = a3;
    }
  } else
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
    float v = (float)acc / cnt;
    // trainer.py:202: torch.nan_to_num(A, nan=0, posinf=0, neginf=0).clamp_min(0)
    if (sanitize) v = (v > 0.0f && v <= 3.402823466e38f) ? v : 0.0f;
    out[((size_t)b * oh + i) * ow + j] = v;
  }
}

// ---- A13 finalize: body attention_maps_finalize_block (profiles_blocks.hpp).  grid = (B, 2) ----
__global__ __launch_bounds__(NT) void attention_maps_finalize_kernel(const PairwisePlan Pw, const PairwisePlan Ph,
                                                                     const MapsFinalizeArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem_d[];
  attention_maps_finalize_block(Pw, Ph, maps_finalize_image(a, blockIdx.x, Pw.nleaves), blockIdx.y, smem_d);
}

template <typename T, typename XF>
static int launch_profiles(const void* A, int B, int H, int W, XF xf, const PairwisePlan& P, double* col, double* ls,
                           hipStream_t st) {
  // one vector load per 4 columns when every strip row starts on a 4-element boundary (leaf offsets are
  // multiples of 8) and the last quad of a row stays inside the image
  const int vec_ok = (W % 4 == 0) && (reinterpret_cast<uintptr_t>(A) % (4 * sizeof(T)) == 0);
  hipLaunchKernelGGL((profiles_kernel<T, XF>), dim3(P.nleaves, B), dim3(NT), 0, st, (const T*)A, H, W, xf, P, col, ls,
                     vec_ok);
  return check_launch("profiles_kernel");
}

}  // namespace attwarp

using namespace attwarp;

extern "C" size_t attwarp_axis_sums_workspace_bytes(int B, int H, int W) {
  PairwisePlan P;
  if (B <= 0 || H <= 0 || !pw_build(W, P)) return 0;
  return (size_t)B * ((size_t)W + (size_t)H * P.nleaves) * sizeof(double);
}

extern "C" int attwarp_gt_marginals(const float* A, int B, int H, int W, float* px, float* py, void* ws,
                                    void* stream) {
  ATTWARP_REQUIRE(A && px && py && ws, "gt_marginals: null pointer");
  ATTWARP_REQUIRE(B > 0 && H > 0 && W > 0, "gt_marginals: non-positive size");
  if (B > 65535) return fail(ATTWARP_E_UNSUPPORTED, "gt_marginals: B > 65535");
  PairwisePlan P;
  if (!pw_build(W, P) || H > 16384) return fail(ATTWARP_E_UNSUPPORTED, "gt_marginals: H, W must be <= 16384");
  double* col = (double*)ws;
  double* ls = col + (size_t)B * W;
  hipStream_t st = as_stream(stream);
  bool handled = false;
  int rc = launch_profiles_f32<XfClampPos>(A, B, H, W, XfClampPos{}, P, col, ls, st, &handled);
  if (!handled) rc = launch_profiles<float, XfClampPos>(A, B, H, W, XfClampPos{}, P, col, ls, st);
  if (rc) return rc;
  const int n = H > W ? H : W;
  hipLaunchKernelGGL(marginals_finalize_kernel, dim3(B, 2), dim3(NT), (size_t)n * sizeof(double), st, col, ls,
                     P.nleaves, H, W, px, py);
  return check_launch("marginals_finalize_kernel");
}

extern "C" int attwarp_adaptive_avg_pool(const float* A, int B, int H, int W, int oh, int ow, int sanitize, float* out,
                                         void* stream) {
  ATTWARP_REQUIRE(A && out, "adaptive_avg_pool: null pointer");
  ATTWARP_REQUIRE(B > 0 && H > 0 && W > 0 && oh > 0 && ow > 0, "adaptive_avg_pool: non-positive size");
  if (oh > 65535 || B > 65535) return fail(ATTWARP_E_UNSUPPORTED, "adaptive_avg_pool: oh/B > 65535");
  if (W > 16384) return fail(ATTWARP_E_UNSUPPORTED, "adaptive_avg_pool: W=%d > 16384", W);
  const int vec_ok = (W % 4 == 0) && (reinterpret_cast<uintptr_t>(A) % 16 == 0);
  hipLaunchKernelGGL(adaptive_pool_kernel, dim3(oh, B), dim3(NT), (size_t)W * sizeof(double), as_stream(stream), A, H,
                     W, oh, ow, sanitize, vec_ok, out);
  return check_launch("adaptive_pool_kernel");
}

// ---- the 256-entry table of the transformed byte values: XfAttention<TR>(0 .. 255), by the device functions the uint8 profile
// kernel uses for its own per-workgroup table (bit-identical), for the one-launch chain steps (chain_step.hip, chain_ragged.hip)
__global__ __launch_bounds__(256) void attention_lut_kernel(int transform, double exp_scale, double exp_divisor, double* __restrict__ lut) {
  const double v = (double)threadIdx.x;
  double r;
  switch (transform) {
    case ATTWARP_T_SQUARE: r = XfAttention<ATTWARP_T_SQUARE>{exp_scale, exp_divisor}(v); break;
    case ATTWARP_T_SQRT: r = XfAttention<ATTWARP_T_SQRT>{exp_scale, exp_divisor}(v); break;
    case ATTWARP_T_EXP: r = XfAttention<ATTWARP_T_EXP>{exp_scale, exp_divisor}(v); break;
    case ATTWARP_T_LOG: r = XfAttention<ATTWARP_T_LOG>{exp_scale, exp_divisor}(v); break;
    default: r = XfAttention<ATTWARP_T_IDENTITY>{exp_scale, exp_divisor}(v); break;
  }
  lut[threadIdx.x] = r;
}

extern "C" int attwarp_attention_transform_lut(int transform, double exp_scale, double exp_divisor, double* lut, void* stream) {
  ATTWARP_REQUIRE(lut, "attention_transform_lut: null pointer");
  ATTWARP_REQUIRE(transform >= ATTWARP_T_IDENTITY && transform <= ATTWARP_T_LOG, "attention_transform_lut: unknown transform %d", transform);
  hipLaunchKernelGGL(attention_lut_kernel, dim3(1), dim3(256), 0, as_stream(stream), transform, exp_scale, exp_divisor, lut);
  return check_launch("attention_lut_kernel");
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
  PairwisePlan Pw, Ph;
  if (!pw_build(w, Pw) || !pw_build(h, Ph))
    return fail(ATTWARP_E_UNSUPPORTED, "axis_maps_from_attention: max(h,w)=%d > 16384", n);
  double* col = (double*)ws;
  double* ls = col + (size_t)B * w;
  hipStream_t st = as_stream(stream);
  int rc = ATTWARP_E_ARG;
#define ATTWARP_PROFILES(TR)                                                                                      \
  case TR: {                                                                                                      \
    XfAttention<TR> xf{exp_scale, exp_divisor};                                                                   \
    if (dtype == ATTWARP_U8) {                                                                                    \
      bool handled = false;                                                                                       \
      rc = launch_profiles_u8<TR>(att, B, h, w, xf, Pw, col, ls, st, &handled);                                   \
      if (!handled) rc = launch_profiles<uint8_t, XfAttention<TR>>(att, B, h, w, xf, Pw, col, ls, st);            \
    }                                                                                                             \
    else if (dtype == ATTWARP_F32) {                                                                              \
      bool handled = false;                                                                                       \
      if (TR == ATTWARP_T_IDENTITY || TR == ATTWARP_T_SQUARE || tune(TUNE_PROFILES_VARIANT) == 2)                 \
        rc = launch_profiles_f32<XfAttention<TR>>(att, B, h, w, xf, Pw, col, ls, st, &handled);                   \
      if (!handled) rc = launch_profiles<float, XfAttention<TR>>(att, B, h, w, xf, Pw, col, ls, st);              \
    }                                                                                                             \
    else rc = launch_profiles<double, XfAttention<TR>>(att, B, h, w, xf, Pw, col, ls, st);                        \
  } break;
  if (dtype != ATTWARP_U8 && dtype != ATTWARP_F32 && dtype != ATTWARP_F64)
    return fail(ATTWARP_E_ARG, "axis_maps_from_attention: dtype must be U8, F32 or F64 (got %d)", dtype);
  switch (transform) {
    ATTWARP_PROFILES(ATTWARP_T_IDENTITY)
    ATTWARP_PROFILES(ATTWARP_T_SQUARE)
    ATTWARP_PROFILES(ATTWARP_T_SQRT)
    ATTWARP_PROFILES(ATTWARP_T_EXP)
    ATTWARP_PROFILES(ATTWARP_T_LOG)
    default: break;
  }
#undef ATTWARP_PROFILES
  if (rc) return rc;
  MapsFinalizeArgs fa{col, ls, h, w, new_w, new_h, transform, exp_scale, exp_divisor, apply_inverse, map_x, map_y, pw_depth(Pw)};
  const size_t lds = maps_finalize_lds_bytes(h, w, Pw, Ph);
  if (const int rc2 = grant_dynamic_lds(attention_maps_finalize_kernel, lds, "axis_maps_from_attention")) return rc2;
  hipLaunchKernelGGL(attention_maps_finalize_kernel, dim3(B, 2), dim3(NT), lds, st, Pw, Ph, fa);
  return check_launch("attention_maps_finalize_kernel");
}
