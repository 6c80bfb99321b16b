// Attention-side stages on the 24x24 token grid:
//   A1  BatchMaskHookLogger._process_attention   AGW/attention_extraction/llava.py:385-396
//   A2  BatchMaskHookLogger.finalize_batch       llava.py:401-411
//   A3  revise_mask (normalize -> enhance -> box) llava.py:207-238
//   A4  ToPILImage (x255, truncate) + PIL resize(LANCZOS) llava.py:192-196,243,253
//
// A1 is the only bandwidth-relevant one: per generation step it reads heads*ntok elements per
// sample (T*32*576*elemsize algorithmic bytes per image over a generation) with
// coalesced row segments; the rest is 576 numbers per image.
//
// Rounding follows the reference's dtype transitions: the hook works in the MODEL dtype
// (float16 in LLaVA), so every reference op boundary rounds to that dtype; reductions are
// accumulated in double and rounded once (see oracle/warp_oracle.py, module docstring).
#include "common.hpp"
#include "attn_f32v.hpp"
#include "mask_blocks.hpp"

namespace attwarp {

constexpr int NT = 256;
constexpr int MAXPL = 16;          // tokens per lane: ntok <= 1024
constexpr int MAX_NTOK = MAXPL * WAVE;

// attention rows are read exactly once per step: nontemporal loads keep them from displacing what the resample kernel
// of the same step left in (and needs from) the L2 / Infinity Cache -- in-step time of this kernel at B=256: 0.110 ->
// 0.074 ms (1024 x 1024 step), 0.112 -> 0.084 ms (336 x 336 step), stand-alone unchanged
template <typename T> __device__ __forceinline__ T nt_load(const T* p);
template <> __device__ __forceinline__ float nt_load<float>(const float* p) { return __builtin_nontemporal_load(p); }
template <> __device__ __forceinline__ __half nt_load<__half>(const __half* p) {
  return __builtin_bit_cast(__half, __builtin_nontemporal_load(reinterpret_cast<const uint16_t*>(p)));
}
template <> __device__ __forceinline__ __hip_bfloat16 nt_load<__hip_bfloat16>(const __hip_bfloat16* p) {
  return __builtin_bit_cast(__hip_bfloat16, __builtin_nontemporal_load(reinterpret_cast<const uint16_t*>(p)));
}

// One workgroup per (pseudo-)sample, any dtype / kv stride / slice length.  Each wave takes heads w, w+4, ...; lane l
// owns tokens l, l+64, ... (NPL = ceil(ntok/64) values per head in registers).  Heads are processed HU at a time so
// that HU*NPL independent coalesced loads are in flight per lane before the first reduction.  The float32 sums follow
// the path's fixed order (attn_f32v.hpp): with one token per lane the four tokens of a quad sit in four neighbouring
// lanes (two xor-shuffles give every lane (x0 + x1) + (x2 + x3)), register i' belongs to 256-token block i'/4 and to
// the lane group r = i' mod 4 of the canonical 64 lanes (canonical lane = 16 r + lane/4), so a lane keeps four running
// sums s_r; the butterfly's first two steps (o = 32, 16) combine the s_r, the last four are shuffles over lane/4.
// grid = nb
template <typename T, int NPL, int HU>
__global__ __launch_bounds__(NT) void attn_reduce_step_kernel(const T* __restrict__ attn, int heads, int64_t sb,
                                                              int64_t sh, int64_t row_off, int64_t skv,
                                                              const int32_t* __restrict__ starts, int starts_mod,
                                                              int max_start, int ntok, T* __restrict__ out) {
  __shared__ float part[NT / WAVE][NPL * WAVE];
  constexpr int NW = NT / WAVE;
  const int b = blockIdx.x;
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  // a slice start outside [0, kv_len - ntok] is clamped for memory safety; the host shims reject such starts
  // (the reference would raise: a truncated slice cannot be stacked, llava.py:390-395)
  const int st = min(max(starts[b % starts_mod], 0), max_start);
  const T* base = attn + (int64_t)b * sb + row_off + (int64_t)st * skv;
  float acc[NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) acc[i] = 0.0f;
  for (int h0 = wid; h0 < heads; h0 += NW * HU) {
    T v[HU][NPL];
#pragma unroll
    for (int u = 0; u < HU; ++u) {
      const int h = h0 + u * NW;
      const T* rp = base + (int64_t)min(h, heads - 1) * sh;      // clamped: tail heads re-read, not used
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const int t = lane + WAVE * i;
        v[u][i] = nt_load<T>(rp + (int64_t)min(t, ntok - 1) * skv);
      }
    }
#pragma unroll
    for (int u = 0; u < HU; ++u) {
      const int h = h0 + u * NW;
      if (h < heads) {                                            // wave uniform
        float sr[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
          const float x = (lane + WAVE * i < ntok) ? to_f32<T>(v[u][i]) : 0.0f;
          const float pr = fadd(x, __shfl_xor(x, 1, WAVE));       // (x0 + x1) in lanes 0,1; (x2 + x3) in lanes 2,3
          const float q = fadd(pr, __shfl_xor(pr, 2, WAVE));      // (x0 + x1) + (x2 + x3) in all four
          sr[i & 3] = fadd(sr[i & 3], q);
        }
        float s = fadd(fadd(sr[0], sr[2]), fadd(sr[1], sr[3]));   // butterfly o = 32, then o = 16
        s = fadd(s, __shfl_xor(s, 32, WAVE));                     // o = 8, 4, 2, 1 over lane / 4
        s = fadd(s, __shfl_xor(s, 16, WAVE));
        s = fadd(s, __shfl_xor(s, 8, WAVE));
        s = fadd(s, __shfl_xor(s, 4, WAVE));
        const T den = add_tiny<T>(from_f32<T>(s));      // (row sum + 1e-12) in the model dtype
#pragma unroll
        for (int i = 0; i < NPL; ++i) acc[i] = fadd(acc[i], to_f32<T>(div_t<T>(v[u][i], den)));
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NPL; ++i) part[wid][lane + WAVE * i] = acc[i];
  __syncthreads();
  const T nheads = from_f32<T>((float)heads);
  for (int t = threadIdx.x; t < ntok; t += NT) {
    float m = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; ++w) m = fadd(m, part[w][t]);
    out[(int64_t)b * ntok + t] = div_t<T>(from_f32<T>(m), nheads);    // mean = sum / N in dtype T
  }
}

// rows with unit kv stride and a slice length that is a multiple of 4: four tokens per lane (body: attn_f32v.hpp)
template <typename T, int NV, int HU>
__global__ __launch_bounds__(NT) void attn_reduce_step_v4_kernel(const AttnStepArgsT<T> a) {
  extern __shared__ __attribute__((aligned(16))) float attn_part[];
  attn_reduce_v4_block<T, NV, HU>(a, blockIdx.x, attn_part);
}

// mean over steps.  steps [Tn, n] -> out [n]
template <typename T>
__global__ __launch_bounds__(NT) void attn_finalize_kernel(const T* __restrict__ steps, int Tn, int64_t n,
                                                           T* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  double acc = 0.0;
  for (int t = 0; t < Tn; ++t) acc += (double)to_f32<T>(steps[(int64_t)t * n + i]);
  out[i] = div_t<T>(from_f64<T>(acc), from_f32<T>((float)Tn));
}

// ---- A3: one workgroup per mask (body: mask_blocks.hpp) ---------------------------------------
__global__ __launch_bounds__(NT) void mask_postproc_kernel(const float* __restrict__ mask, int n, int ks, float coe,
                                                           float* __restrict__ out) {
  __shared__ float x[32 * 32];
  __shared__ double red[NT / WAVE];
  __shared__ float fred[2][NT / WAVE];
  mask_postproc_block(mask, n, ks, coe, out, blockIdx.x, x, red, &fred[0][0]);
}

// ---- A4: Pillow's 8-bit separable resampler (ImagingResampleHorizontal/Vertical_8bpc); helpers in mask_blocks.hpp ----
// horizontal pass: tmp[b][y][xx], one thread per output.  grid = (ceil(out_w/NT), h, B)
__global__ __launch_bounds__(NT) void lanczos_h_kernel(const float* __restrict__ mf, const uint8_t* __restrict__ mu,
                                                       int h, int w, int out_w, const int32_t* __restrict__ bounds,
                                                       const int32_t* __restrict__ kk, int ksize,
                                                       uint8_t* __restrict__ tmp) {
  const int xx = blockIdx.x * NT + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (xx >= out_w) return;
  const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
  const int32_t* k = kk + (size_t)xx * ksize;
  const size_t row = ((size_t)b * h + y) * w;
  int ss = 1 << (PIL_PRECISION_BITS - 1);
  for (int x = 0; x < cnt; ++x) {
    const int px = mf ? (int)to_pil_u8(mf[row + xmin + x]) : (int)mu[row + xmin + x];
    ss += px * k[x];
  }
  tmp[((size_t)b * h + y) * out_w + xx] = pil_clip8(ss);
}

// vertical pass.  grid = (ceil(out_w/NT), out_h, B)
__global__ __launch_bounds__(NT) void lanczos_v_kernel(const uint8_t* __restrict__ tmp, int h, int out_w, int out_h,
                                                       const int32_t* __restrict__ bounds,
                                                       const int32_t* __restrict__ kk, int ksize,
                                                       uint8_t* __restrict__ out) {
  const int xx = blockIdx.x * NT + threadIdx.x, yy = blockIdx.y, b = blockIdx.z;
  if (xx >= out_w) return;
  const int ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
  const int32_t* k = kk + (size_t)yy * ksize;
  int ss = 1 << (PIL_PRECISION_BITS - 1);
  for (int y = 0; y < cnt; ++y) ss += (int)tmp[((size_t)b * h + ymin + y) * out_w + xx] * k[y];
  out[((size_t)b * out_h + yy) * out_w + xx] = pil_clip8(ss);
}

// vertical pass, 4 pixels per thread (out_w % 4 == 0): 4-byte loads and stores, 256 B per wave instruction.
// grid = (ceil(out_w/4/NT), out_h, B)
__global__ __launch_bounds__(NT) void lanczos_v4_kernel(const uint8_t* __restrict__ tmp, int h, int out_w, int out_h,
                                                        const int32_t* __restrict__ bounds,
                                                        const int32_t* __restrict__ kk, int ksize,
                                                        uint8_t* __restrict__ out) {
  const int x4 = blockIdx.x * NT + threadIdx.x, yy = blockIdx.y, b = blockIdx.z;
  if (x4 * 4 >= out_w) return;
  const int ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
  const int32_t* k = kk + (size_t)yy * ksize;
  int s0 = 1 << (PIL_PRECISION_BITS - 1), s1 = s0, s2 = s0, s3 = s0;
  const uint8_t* col = tmp + ((size_t)b * h + ymin) * out_w + (size_t)x4 * 4;
  for (int y = 0; y < cnt; ++y) {
    const uchar4 p = *reinterpret_cast<const uchar4*>(col + (size_t)y * out_w);
    const int w = k[y];
    s0 += (int)p.x * w; s1 += (int)p.y * w; s2 += (int)p.z * w; s3 += (int)p.w * w;
  }
  uchar4 o;
  o.x = pil_clip8(s0); o.y = pil_clip8(s1); o.z = pil_clip8(s2); o.w = pil_clip8(s3);
  *reinterpret_cast<uchar4*>(out + ((size_t)b * out_h + yy) * out_w + (size_t)x4 * 4) = o;
}

// Both passes in one launch for SMALL sources (the 24 x 24 token grid): the workgroup of (image, block of R output
// rows) quantises the source rows it needs into LDS, runs the horizontal pass for them into an LDS uint8 tile
// [h][out_w] and the vertical pass from that tile, 4 output pixels per thread (one ds_read_b32 per tap row, one
// dword store).  Same integer arithmetic as the two-kernel form; no tmp round trip, one launch instead of two.
// grid = (ceil(out_h / R), B); out_w % 4 == 0; ksize_x <= KS.  LDS: src[h*w] | tile[h*out_w]
template <int KS>
__global__ __launch_bounds__(NT) void lanczos_fused_kernel(const float* __restrict__ mf, const uint8_t* __restrict__ mu,
                                                           int h, int w, int out_h, int out_w,
                                                           const int32_t* __restrict__ bounds_x,
                                                           const int32_t* __restrict__ kk_x, int ksize_x,
                                                           const int32_t* __restrict__ bounds_y,
                                                           const int32_t* __restrict__ kk_y, int ksize_y, int R,
                                                           uint8_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lz[];
  __shared__ int32_t s_ky[64 * 32];                  // vertical coefficients and bounds of the block's rows (R <= 64)
  __shared__ int32_t s_by[64 * 2];
  const int srcp = (h * w + 3) & ~3;
  uint8_t* src = lz;
  uint8_t* tile = lz + srcp;
  const int tid = threadIdx.x, b = blockIdx.y;
  const int yy0 = blockIdx.x * R, yy1 = min(yy0 + R, out_h), nrows = yy1 - yy0;
  const int ys = bounds_y[2 * yy0];                                          // first / one-past-last source row
  const int ye = bounds_y[2 * (yy1 - 1)] + bounds_y[2 * (yy1 - 1) + 1];
  const size_t ib = (size_t)b * h * w;
  for (int i = ys * w + tid; i < ye * w; i += NT) src[i] = mf ? to_pil_u8(mf[ib + i]) : mu[ib + i];
  if (ksize_y != 8) {      // only the generic vertical pass reads these (R <= 64 there)
    for (int i = tid; i < nrows * ksize_y; i += NT) s_ky[i] = kk_y[(size_t)yy0 * ksize_y + i];
    if (tid < 2 * nrows) s_by[tid] = bounds_y[2 * yy0 + tid];
  }
  __syncthreads();
  for (int xx = tid; xx < out_w; xx += NT) {
    const int xmin = bounds_x[2 * xx], cnt = bounds_x[2 * xx + 1];
    int kreg[KS], toff[KS];
#pragma unroll
    for (int x = 0; x < KS; ++x) {
      kreg[x] = (x < cnt) ? kk_x[(size_t)xx * ksize_x + min(x, ksize_x - 1)] : 0;
      toff[x] = xmin + min(x, cnt - 1);
    }
    for (int r = ys; r < ye; ++r) {
      const uint8_t* row = src + r * w;
      int ss = 1 << (PIL_PRECISION_BITS - 1);
#pragma unroll
      for (int x = 0; x < KS; ++x) ss += __mul24((int)row[toff[x]], kreg[x]);
      tile[r * out_w + xx] = pil_clip8(ss);
    }
  }
  __syncthreads();
  // vertical pass.  Everything that depends on the output row only (bounds, the <= 8 coefficients) is wave uniform
  // and comes from scalar loads; a lane owns one dword column (4 pixels) and keeps the 8 tile rows of the current tap
  // window UNPACKED in registers, so an output dword costs 32 v_mad_i32_i24 (pixels are 8 bit, coefficients < 2^23)
  // + clip / pack / one store.  The window slides by one tile row every out_h / h output rows (42 for 24 -> 1024).
  // Work items = (64-dword column chunk, chunk of the block's rows), dealt to the 4 waves.
  const int nq = out_w >> 2;
  if (ksize_y == 8) {       // rows of exactly 8 coefficients: 32-byte scalar loads, masked by the row's tap count
    const int lane = tid & (WAVE - 1);
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int QW = (nq + WAVE - 1) / WAVE;
    const int RC = QW >= 3 ? 1 : (NT / WAVE) / QW;               // QW = 1 -> 4 row chunks, 2 -> 2
    const int rows_per = (nrows + RC - 1) / RC;
    for (int item = wid; item < QW * RC; item += NT / WAVE) {
      const int rc = item / QW, qc = item - rc * QW;
      const int q = qc * WAVE + lane;
      const int qq = min(q, nq - 1);
      const int r_beg = rc * rows_per, r_end = min(r_beg + rows_per, nrows);
      int win[8][4];
      int base = -0x40000000;
#pragma unroll
      for (int y = 0; y < 8; ++y) win[y][0] = win[y][1] = win[y][2] = win[y][3] = 0;
      auto load_row = [&](int r, int (&dst)[4]) {
        const uint32_t wv = reinterpret_cast<const uint32_t*>(tile + min(r, h - 1) * out_w)[qq];
        dst[0] = (int)(wv & 0xffu); dst[1] = (int)((wv >> 8) & 0xffu); dst[2] = (int)((wv >> 16) & 0xffu); dst[3] = (int)(wv >> 24);
      };
      // the NEXT row's bounds and coefficients are fetched while the current row is computed
      // (coefficients past the row's tap count are masked here, so the caller's table need not be zero-padded)
      int kc[8], ymin_c = 0;
      if (r_beg < r_end) {
        ymin_c = bounds_y[2 * (yy0 + r_beg)];
        const int cnt_c = bounds_y[2 * (yy0 + r_beg) + 1];
#pragma unroll
        for (int y = 0; y < 8; ++y) kc[y] = y < cnt_c ? kk_y[(size_t)(yy0 + r_beg) * 8 + y] : 0;
      }
      for (int ry = r_beg; ry < r_end; ++ry) {
        const int yy = yy0 + ry;
        const int ymin = ymin_c;
        int kv[8];
#pragma unroll
        for (int y = 0; y < 8; ++y) kv[y] = kc[y];
        const int yn = min(yy + 1, out_h - 1);
        ymin_c = bounds_y[2 * yn];
        const int cnt_n = bounds_y[2 * yn + 1];
#pragma unroll
        for (int y = 0; y < 8; ++y) kc[y] = y < cnt_n ? kk_y[(size_t)yn * 8 + y] : 0;
        if (ymin != base) {                                      // wave uniform
          if (ymin == base + 1) {
#pragma unroll
            for (int y = 0; y < 7; ++y) { win[y][0] = win[y + 1][0]; win[y][1] = win[y + 1][1]; win[y][2] = win[y + 1][2]; win[y][3] = win[y + 1][3]; }
            load_row(ymin + 7, win[7]);
          } else {
#pragma unroll
            for (int y = 0; y < 8; ++y) load_row(ymin + y, win[y]);
          }
          base = ymin;
        }
        int s0 = 1 << (PIL_PRECISION_BITS - 1), s1 = s0, s2 = s0, s3 = s0;
#pragma unroll
        for (int y = 0; y < 8; ++y) {
          // one v_mad_i32_i24 per tap and pixel (left to itself the compiler pairs v_mul_i32_i24 with v_add3_u32:
          // 1.5 instructions per tap); the coefficient is wave uniform and sits in an SGPR
          asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(s0) : "v"(win[y][0]), "s"(kv[y]));
          asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(s1) : "v"(win[y][1]), "s"(kv[y]));
          asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(s2) : "v"(win[y][2]), "s"(kv[y]));
          asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(s3) : "v"(win[y][3]), "s"(kv[y]));
        }
        if (q < nq) reinterpret_cast<uint32_t*>(out + ((size_t)b * out_h + yy) * out_w)[q] = pil_clip8x4(s0, s1, s2, s3);
      }
    }
    return;
  }
  // generic form (more than 8 vertical taps: down-sampling): a thread owns dword q of an output row; rows are spread
  // over the threads left
  const int rstep = nq >= NT ? 1 : NT / nq;          // rows in flight per sweep
  const int q0 = nq >= NT ? tid : tid % nq, r0 = nq >= NT ? 0 : tid / nq;
  if (r0 < rstep) {
    for (int ry = r0; ry < nrows; ry += rstep) {
      const int yy = yy0 + ry;
      const int ymin = s_by[2 * ry], cnt = s_by[2 * ry + 1];
      const int32_t* k = s_ky + ry * ksize_y;
      for (int q = q0; q < nq; q += NT) {
        int s0 = 1 << (PIL_PRECISION_BITS - 1), s1 = s0, s2 = s0, s3 = s0;
        for (int y = 0; y < cnt; ++y) {
          const uint32_t wv = reinterpret_cast<const uint32_t*>(tile + (ymin + y) * out_w)[q];
          const int kv = k[y];
          s0 += (int)(wv & 0xffu) * kv;
          s1 += (int)((wv >> 8) & 0xffu) * kv;
          s2 += (int)((wv >> 16) & 0xffu) * kv;
          s3 += (int)(wv >> 24) * kv;
        }
        reinterpret_cast<uint32_t*>(out + ((size_t)b * out_h + yy) * out_w)[q] = pil_clip8x4(s0, s1, s2, s3);
      }
    }
  }
}

// Up-sampling form of the fused kernel (<= 8 taps on both axes): body lanczos_strip_block (mask_blocks.hpp).
// grid = (nstrips * nchunks, B).  LDS: src[h*w] | tile[h][256].   kk_y rows hold exactly 8 coefficients (zero padded).
template <int KS, bool UA>
__global__ __launch_bounds__(NT) void lanczos_strip_kernel(const LanczosStripArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lz[];
  const int b = blockIdx.y;
  lanczos_strip_block<KS, UA>(a, blockIdx.x, (size_t)b * a.h * a.w, a.out + (size_t)b * a.out_h * a.out_w, lz);
}

// copy / quantise pass used when an axis keeps its size (Pillow skips that pass)
__global__ __launch_bounds__(NT) void quantise_copy_kernel(const float* __restrict__ mf, const uint8_t* __restrict__ mu,
                                                           size_t n, uint8_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
  if (i < n) out[i] = mf ? to_pil_u8(mf[i]) : mu[i];
}

// ---- "next" row 3: warped uint8 RGB image -> CLIP-ready tensor (HF CLIPImageProcessor, PIL backend) ----
// horizontal pass of Pillow's 8-bit resampler on interleaved C-channel images, restricted to the output
// columns [left, left+ow):  tmp[b][y][xx][c].   grid = (ceil(ow*C/NT), h, B)
__global__ __launch_bounds__(NT) void resample8_h_kernel(const uint8_t* __restrict__ src, int h, int w, int C,
                                                         int left, int ow, const int32_t* __restrict__ bounds,
                                                         const int32_t* __restrict__ kk, int ksize,
                                                         uint8_t* __restrict__ tmp) {
  const int e = blockIdx.x * NT + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (e >= ow * C) return;
  const int xx = e / C, c = e - xx * C;
  const int xo = left + xx;
  const int xmin = bounds[2 * xo], cnt = bounds[2 * xo + 1];
  const int32_t* k = kk + (size_t)xo * ksize;
  const uint8_t* row = src + ((size_t)b * h + y) * w * C + c;
  int ss = 1 << (PIL_PRECISION_BITS - 1);
  for (int x = 0; x < cnt; ++x) ss += (int)row[(size_t)(xmin + x) * C] * k[x];
  tmp[((size_t)b * h + y) * ow * C + e] = pil_clip8(ss);
}

// vertical pass for the output rows [top, top+oh) fused with the CLIP epilogue:
//   u8 -> float32(float64(u8) * (1/255)) -> (x - mean[c]) / std[c] in float32 -> planar [B,C,oh,ow] (f32 or f16)
// grid = (ceil(ow/NT), oh, B)
template <typename OutT>
__global__ __launch_bounds__(NT) void clip_v_kernel(const uint8_t* __restrict__ tmp, int h, int C, int top, int oh,
                                                    int ow, const int32_t* __restrict__ bounds,
                                                    const int32_t* __restrict__ kk, int ksize, float m0, float m1,
                                                    float m2, float m3, float s0, float s1, float s2, float s3,
                                                    OutT* __restrict__ out) {
  const int xx = blockIdx.x * NT + threadIdx.x, yy = blockIdx.y, b = blockIdx.z;
  if (xx >= ow) return;
  const int yo = top + yy;
  const int ymin = bounds[2 * yo], cnt = bounds[2 * yo + 1];
  const int32_t* k = kk + (size_t)yo * ksize;
  const float mean[4] = {m0, m1, m2, m3}, stdv[4] = {s0, s1, s2, s3};
  for (int c = 0; c < C; ++c) {
    int ss = 1 << (PIL_PRECISION_BITS - 1);
    for (int y = 0; y < cnt; ++y) ss += (int)tmp[(((size_t)b * h + ymin + y) * ow + xx) * C + c] * k[y];
    const float x = (float)((double)pil_clip8(ss) * (1.0 / 255.0));
    const float v = fsub(x, mean[c]) / stdv[c];
    out[(((size_t)b * C + c) * oh + yy) * ow + xx] = from_f32<OutT>(v);
  }
}

template <typename T>
static int launch_step(const void* attn, int nb, int heads, int64_t sb, int64_t sh, int64_t row_off, int64_t skv,
                       const int32_t* starts, int starts_mod, int max_start, int ntok, void* out, hipStream_t st) {
  // contiguous kv, slice length a multiple of 4 and <= 768: four tokens per lane (16-byte / 8-byte loads)
  if (skv == 1 && ntok % 4 == 0 && ntok >= 4 && ntok <= 3 * 4 * WAVE) {
    AttnStepArgsT<T> a;
    a.attn = (const T*)attn; a.heads = heads; a.sb = sb; a.sh = sh; a.row_off = row_off; a.starts = starts;
    a.starts_mod = starts_mod; a.max_start = max_start; a.ntok = ntok; a.out = (T*)out;
    const int hu = tune(TUNE_ATTN_HU);
    if (hu == 2)
      hipLaunchKernelGGL((attn_reduce_step_v4_kernel<T, 3, 2>), dim3(nb), dim3(NT), attn_v4_lds_bytes<3>(), st, a);
    else if (hu == 8)
      hipLaunchKernelGGL((attn_reduce_step_v4_kernel<T, 3, 8>), dim3(nb), dim3(NT), attn_v4_lds_bytes<3>(), st, a);
    else if (hu == 1)
      hipLaunchKernelGGL((attn_reduce_step_v4_kernel<T, 3, 1>), dim3(nb), dim3(NT), attn_v4_lds_bytes<3>(), st, a);
    else
      hipLaunchKernelGGL((attn_reduce_step_v4_kernel<T, 3, 4>), dim3(nb), dim3(NT), attn_v4_lds_bytes<3>(), st, a);
    return check_launch("attn_reduce_step_v4_kernel");
  }
  if (ntok <= 9 * WAVE)      // 576 image tokens (LLaVA-1.5): 9 per lane, 4 heads in flight
    hipLaunchKernelGGL((attn_reduce_step_kernel<T, 9, 4>), dim3(nb), dim3(NT), 0, st, (const T*)attn, heads, sb, sh,
                       row_off, skv, starts, starts_mod, max_start, ntok, (T*)out);
  else
    hipLaunchKernelGGL((attn_reduce_step_kernel<T, MAXPL, 2>), dim3(nb), dim3(NT), 0, st, (const T*)attn, heads, sb,
                       sh, row_off, skv, starts, starts_mod, max_start, ntok, (T*)out);
  return check_launch("attn_reduce_step_kernel");
}
template <typename T>
static int launch_finalize(const void* steps, int Tn, int64_t n, void* out, hipStream_t st) {
  hipLaunchKernelGGL((attn_finalize_kernel<T>), dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, st, (const T*)steps,
                     Tn, n, (T*)out);
  return check_launch("attn_finalize_kernel");
}

// dtype-dispatched A1 launch for other translation units (probe.hip chains into it)
int launch_attn_step_dtype(int dtype, const void* attn, int nb, int heads, int64_t sb, int64_t sh, int64_t row_off,
                           int64_t skv, const int32_t* starts, int starts_mod, int max_start, int ntok, void* out,
                           hipStream_t st) {
  switch (dtype) {
    case ATTWARP_F32: return launch_step<float>(attn, nb, heads, sb, sh, row_off, skv, starts, starts_mod, max_start, ntok, out, st);
    case ATTWARP_F16: return launch_step<__half>(attn, nb, heads, sb, sh, row_off, skv, starts, starts_mod, max_start, ntok, out, st);
    default: return launch_step<__hip_bfloat16>(attn, nb, heads, sb, sh, row_off, skv, starts, starts_mod, max_start, ntok, out, st);
  }
}

static int check_attn_dtype(int dtype, const char* who) {
  if (dtype == ATTWARP_F32 || dtype == ATTWARP_F16 || dtype == ATTWARP_BF16) return 0;
  return fail(ATTWARP_E_ARG, "%s: dtype must be F32, F16 or BF16 (got %d)", who, dtype);
}

}  // namespace attwarp

using namespace attwarp;

extern "C" int attwarp_attn_reduce_step(const void* attn, int dtype, int B, int heads, int q_len, int kv_len,
                                        int64_t stride_b, int64_t stride_h, int64_t stride_q, int64_t stride_kv,
                                        const int32_t* starts, int ntok, void* out, void* stream) {
  ATTWARP_REQUIRE(attn && starts && out, "attn_reduce_step: null pointer");
  ATTWARP_REQUIRE(B > 0 && heads > 0 && q_len > 0 && kv_len > 0 && ntok > 0, "attn_reduce_step: non-positive size");
  ATTWARP_REQUIRE(ntok <= kv_len, "attn_reduce_step: ntok=%d > kv_len=%d", ntok, kv_len);
  if (check_attn_dtype(dtype, "attn_reduce_step")) return ATTWARP_E_ARG;
  if (ntok > MAX_NTOK) return fail(ATTWARP_E_UNSUPPORTED, "attn_reduce_step: ntok=%d > %d", ntok, MAX_NTOK);
  const int64_t row_off = (int64_t)(q_len - 1) * stride_q;
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case ATTWARP_F32: return launch_step<float>(attn, B, heads, stride_b, stride_h, row_off, stride_kv, starts, B, kv_len - ntok, ntok, out, st);
    case ATTWARP_F16: return launch_step<__half>(attn, B, heads, stride_b, stride_h, row_off, stride_kv, starts, B, kv_len - ntok, ntok, out, st);
    default: return launch_step<__hip_bfloat16>(attn, B, heads, stride_b, stride_h, row_off, stride_kv, starts, B, kv_len - ntok, ntok, out, st);
  }
}

extern "C" int attwarp_attn_finalize(const void* steps, int dtype, int T, int B, int ntok, void* out, void* stream) {
  ATTWARP_REQUIRE(steps && out, "attn_finalize: null pointer");
  ATTWARP_REQUIRE(T > 0 && B > 0 && ntok > 0, "attn_finalize: non-positive size");
  if (check_attn_dtype(dtype, "attn_finalize")) return ATTWARP_E_ARG;
  const int64_t n = (int64_t)B * ntok;
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case ATTWARP_F32: return launch_finalize<float>(steps, T, n, out, st);
    case ATTWARP_F16: return launch_finalize<__half>(steps, T, n, out, st);
    default: return launch_finalize<__hip_bfloat16>(steps, T, n, out, st);
  }
}

extern "C" size_t attwarp_attn_reduce_stack_workspace_bytes(int dtype, int T, int B, int ntok) {
  if (T <= 0 || B <= 0 || ntok <= 0) return 0;
  return (size_t)T * B * ntok * (dtype == ATTWARP_F32 ? 4 : 2);
}

extern "C" int attwarp_attn_reduce_stack(const void* rows, int dtype, int T, int B, int heads, int kv_len,
                                         const int32_t* starts, int ntok, void* out, void* ws, void* stream) {
  ATTWARP_REQUIRE(rows && starts && out && ws, "attn_reduce_stack: null pointer");
  ATTWARP_REQUIRE(T > 0 && B > 0 && heads > 0 && kv_len > 0 && ntok > 0, "attn_reduce_stack: non-positive size");
  ATTWARP_REQUIRE(ntok <= kv_len, "attn_reduce_stack: ntok=%d > kv_len=%d", ntok, kv_len);
  if (check_attn_dtype(dtype, "attn_reduce_stack")) return ATTWARP_E_ARG;
  if (ntok > MAX_NTOK) return fail(ATTWARP_E_UNSUPPORTED, "attn_reduce_stack: ntok=%d > %d", ntok, MAX_NTOK);
  hipStream_t st = as_stream(stream);
  const int64_t sb = (int64_t)heads * kv_len, sh = kv_len;
  int rc;
  switch (dtype) {
    case ATTWARP_F32: rc = launch_step<float>(rows, T * B, heads, sb, sh, 0, 1, starts, B, kv_len - ntok, ntok, ws, st); break;
    case ATTWARP_F16: rc = launch_step<__half>(rows, T * B, heads, sb, sh, 0, 1, starts, B, kv_len - ntok, ntok, ws, st); break;
    default: rc = launch_step<__hip_bfloat16>(rows, T * B, heads, sb, sh, 0, 1, starts, B, kv_len - ntok, ntok, ws, st); break;
  }
  if (rc) return rc;
  return attwarp_attn_finalize(ws, dtype, T, B, ntok, out, stream);
}

extern "C" int attwarp_mask_postproc(const float* mask, int B, int n, int kernel_size, float enhance_coe, float* out,
                                     void* stream) {
  ATTWARP_REQUIRE(mask && out, "mask_postproc: null pointer");
  ATTWARP_REQUIRE(B > 0 && n > 0, "mask_postproc: non-positive size");
  ATTWARP_REQUIRE(kernel_size > 0 && (kernel_size & 1), "mask_postproc: kernel_size must be odd (got %d)", kernel_size);
  if (n > 32 || kernel_size > 7) return fail(ATTWARP_E_UNSUPPORTED, "mask_postproc: n <= 32 and kernel_size <= 7");
  hipLaunchKernelGGL(mask_postproc_kernel, dim3(B), dim3(NT), 0, as_stream(stream), mask, n, kernel_size, enhance_coe,
                     out);
  return check_launch("mask_postproc_kernel");
}

extern "C" int attwarp_mask_upsample_lanczos(const float* mask_f32, const uint8_t* mask_u8, int B, int h, int w,
                                             int out_h, int out_w, const int32_t* bounds_x, const int32_t* kk_x,
                                             int ksize_x, const int32_t* bounds_y, const int32_t* kk_y, int ksize_y,
                                             uint8_t* tmp, uint8_t* out, void* stream) {
  ATTWARP_REQUIRE((mask_f32 != nullptr) != (mask_u8 != nullptr), "mask_upsample_lanczos: pass exactly one of mask_f32 / mask_u8");
  ATTWARP_REQUIRE(out, "mask_upsample_lanczos: null output");
  ATTWARP_REQUIRE(B > 0 && h > 0 && w > 0 && out_h > 0 && out_w > 0, "mask_upsample_lanczos: non-positive size");
  if (B > 65535 || h > 65535 || out_h > 65535) return fail(ATTWARP_E_UNSUPPORTED, "mask_upsample_lanczos: dims > 65535");
  const bool need_h = out_w != w, need_v = out_h != h;
  ATTWARP_REQUIRE(!need_h || (bounds_x && kk_x && ksize_x > 0), "mask_upsample_lanczos: missing x coefficients");
  ATTWARP_REQUIRE(!need_v || (bounds_y && kk_y && ksize_y > 0), "mask_upsample_lanczos: missing y coefficients");
  ATTWARP_REQUIRE(!(need_h && need_v) || tmp, "mask_upsample_lanczos: tmp workspace required for a two-pass resize");
  hipStream_t st = as_stream(stream);
  if (!need_h && !need_v) {
    const size_t n = (size_t)B * h * w;
    hipLaunchKernelGGL(quantise_copy_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, st, mask_f32, mask_u8, n, out);
    return check_launch("quantise_copy_kernel");
  }
  // up-sampling of a small source with <= 8 taps per axis (the 24 x 24 token grid): column-strip kernel
  if (need_h && need_v && tune(TUNE_LANCZOS_VARIANT) <= 0 && (long long)h * w <= 4096 && ksize_x <= 8 &&
      ksize_y == 8 && (size_t)((h * w + 3) & ~3) + (size_t)h * NT <= 48 * 1024) {
    // rows that do not start on dword boundaries (out_w % 4 != 0, e.g. the mask of a 683-pixel-wide image): the UA form
    const bool ua = out_w % 4 != 0 || (reinterpret_cast<uintptr_t>(out) & 3u) != 0;
    const int nstrips = (out_w + NT - 1) / NT;
    // enough workgroups to fill the chip (~4096: measured 1024x1024 B=256, rows per chunk 512 / 256 / 128 / 64:
    // 108 / 105 / 111 / 127 us), at least 16 rows per wave
    long long nchunks = (4096 + (long long)B * nstrips - 1) / ((long long)B * nstrips);
    const int max_chunks = (out_h + 63) / 64;
    if (nchunks > max_chunks) nchunks = max_chunks;
    if (nchunks < 1) nchunks = 1;
    int rows_per_chunk = (int)((out_h + nchunks - 1) / nchunks);
    if (const int v = tune(TUNE_LANCZOS_ROWS); v >= 1) rows_per_chunk = v < out_h ? v : out_h;
    nchunks = (out_h + rows_per_chunk - 1) / rows_per_chunk;
    const size_t lds = (size_t)((h * w + 3) & ~3) + (size_t)h * NT;
    LanczosStripArgs la{mask_f32, mask_u8, h, w, out_h, out_w, bounds_x, kk_x, ksize_x, bounds_y, kk_y, (int)nchunks, rows_per_chunk, out};
    if (ua) hipLaunchKernelGGL((lanczos_strip_kernel<8, true>), dim3((unsigned)(nstrips * nchunks), B), dim3(NT), lds, st, la);
    else hipLaunchKernelGGL((lanczos_strip_kernel<8, false>), dim3((unsigned)(nstrips * nchunks), B), dim3(NT), lds, st, la);
    return check_launch("lanczos_strip_kernel");
  }
  // small source and both passes needed, any tap count: one fused launch (row-block form)
  {
    const bool two_kernel = tune(TUNE_LANCZOS_VARIANT) == 1;      // 2: the row-block fused kernel
    const size_t lds = (size_t)((h * w + 3) & ~3) + (size_t)h * out_w;
    if (need_h && need_v && !two_kernel && (long long)h * w <= 4096 && out_w % 4 == 0 && ksize_x <= 8 &&
        ksize_y <= 32 &&
        lds <= 48 * 1024 && (reinterpret_cast<uintptr_t>(out) & 3u) == 0) {
      int R = out_h >= 448 ? 64 : 32;   // measured: 24->336 R=32 27 us (two-kernel form 34), 24->1024 B=256 R=64 258 us (287)
      if (const int v = tune(TUNE_LANCZOS_ROWS); v >= 1 && v <= (ksize_y == 8 ? 1024 : 64)) R = v;
      hipLaunchKernelGGL((lanczos_fused_kernel<8>), dim3((out_h + R - 1) / R, B), dim3(NT), lds, st, mask_f32, mask_u8,
                         h, w, out_h, out_w, bounds_x, kk_x, ksize_x, bounds_y, kk_y, ksize_y, R, out);
      return check_launch("lanczos_fused_kernel");
    }
  }
  const uint8_t* vsrc = mask_u8;
  if (need_h) {
    uint8_t* hdst = need_v ? tmp : out;
    hipLaunchKernelGGL(lanczos_h_kernel, dim3((out_w + NT - 1) / NT, h, B), dim3(NT), 0, st, mask_f32, mask_u8, h, w,
                       out_w, bounds_x, kk_x, ksize_x, hdst);
    int rc = check_launch("lanczos_h_kernel");
    if (rc) return rc;
    vsrc = hdst;
  } else if (mask_f32) {
    ATTWARP_REQUIRE(tmp, "mask_upsample_lanczos: tmp workspace required to quantise a float mask");
    const size_t n = (size_t)B * h * w;
    hipLaunchKernelGGL(quantise_copy_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, st, mask_f32, nullptr, n, tmp);
    int rc = check_launch("quantise_copy_kernel");
    if (rc) return rc;
    vsrc = tmp;
  }
  if (need_v) {
    const bool vec4 = (out_w % 4 == 0) && ((reinterpret_cast<uintptr_t>(vsrc) | reinterpret_cast<uintptr_t>(out)) % 4 == 0);
    if (vec4)
      hipLaunchKernelGGL(lanczos_v4_kernel, dim3((out_w / 4 + NT - 1) / NT, out_h, B), dim3(NT), 0, st, vsrc, h, out_w,
                         out_h, bounds_y, kk_y, ksize_y, out);
    else
      hipLaunchKernelGGL(lanczos_v_kernel, dim3((out_w + NT - 1) / NT, out_h, B), dim3(NT), 0, st, vsrc, h, out_w,
                         out_h, bounds_y, kk_y, ksize_y, out);
    return check_launch("lanczos_v_kernel");
  }
  return ATTWARP_OK;
}

namespace attwarp {
// Generic two-kernel form (one thread per output element, global byte taps): any size; clip.hip holds the
// LDS-staged kernels that take the common shapes and calls this one otherwise.
int clip_preprocess_generic(const uint8_t* src, int B, int h, int w, int C, int top, int left, int size,
                            const int32_t* bounds_x, const int32_t* kk_x, int ksize_x, const int32_t* bounds_y,
                            const int32_t* kk_y, int ksize_y, const float* m, const float* s, uint8_t* tmp, void* out,
                            int out_dtype, hipStream_t st) {
  hipLaunchKernelGGL(resample8_h_kernel, dim3((size * C + NT - 1) / NT, h, B), dim3(NT), 0, st, src, h, w, C, left, size,
                     bounds_x, kk_x, ksize_x, tmp);
  int rc = check_launch("resample8_h_kernel");
  if (rc) return rc;
  const dim3 grid((size + NT - 1) / NT, size, B);
  if (out_dtype == ATTWARP_F32)
    hipLaunchKernelGGL((clip_v_kernel<float>), grid, dim3(NT), 0, st, tmp, h, C, top, size, size, bounds_y, kk_y,
                       ksize_y, m[0], m[1], m[2], m[3], s[0], s[1], s[2], s[3], (float*)out);
  else
    hipLaunchKernelGGL((clip_v_kernel<__half>), grid, dim3(NT), 0, st, tmp, h, C, top, size, size, bounds_y, kk_y,
                       ksize_y, m[0], m[1], m[2], m[3], s[0], s[1], s[2], s[3], (__half*)out);
  return check_launch("clip_v_kernel");
}
}  // namespace attwarp
