// "next" row 3: warped uint8 image -> CLIP-ready tensor, replacing the PNG round trip
// AGW/new_method.py:491 (cv2.imwrite) -> AGW/evaluate_accuracy.py:157-158 (Image.open, process_images).
// Arithmetic = HF CLIPImageProcessor (PIL backend) as LLaVA-1.5 configures it: Pillow's 8-bit two-pass BICUBIC
// resample (22-bit fixed-point coefficients, uint8 between the passes), centre crop, float32(float64(u8) * (1/255)),
// (x - mean) / std in float32, channels first.  The coefficient tables are Pillow's, computed on the host
// (attwarp_amd/_tables.py) and shared by every image of the batch.
//
// Two LDS-staged kernels (the generic one-thread-per-output form in attn.hip costs 0.20 + 0.12 ms for 64 images
// 500x500 -> 336x336, bound by global byte loads: 14 and 8 load instructions per output byte):
//   clip_h_rows_kernel : a workgroup stages the coefficient table of the cropped columns and RB source rows in LDS
//                        (coalesced dword loads), computes RB x ow pixels from LDS, writes the uint8 rows back
//                        with dword stores.
//   clip_v_rows_kernel : a workgroup builds the 256-entry-per-channel table of the epilogue (the float64 multiply
//                        and the IEEE division happen 768 times per workgroup instead of once per output), then
//                        each thread accumulates 4 interleaved bytes per tap row from one dword load, looks the
//                        result up and the row is written planar through an LDS transpose.
#include "common.hpp"

namespace attwarp {

int clip_preprocess_generic(const uint8_t* src, int B, int h, int w, int C, int top, int left, int size,
                            const int32_t* bounds_x, const int32_t* kk_x, int ksize_x, const int32_t* bounds_y,
                            const int32_t* kk_y, int ksize_y, const float* m, const float* s, uint8_t* tmp, void* out,
                            int out_dtype, hipStream_t st);

namespace clip {

constexpr int NT = 256;
constexpr int PIL_PRECISION_BITS = 32 - 8 - 2;
constexpr int RV = 8;            // output rows per workgroup of the vertical pass (the per-workgroup table and index set-up is amortised over them)

__device__ __forceinline__ uint8_t clip8(int v) { return (uint8_t)min(max(v >> PIL_PRECISION_BITS, 0), 255); }
// acc += pixel * coefficient with ONE full-rate instruction: pixels are 8 bit, Pillow's coefficients |k| <= 2^22, so the
// 24-bit multiply-add is exact (left to itself the compiler emits the quarter-rate v_mul_lo_u32 + an add: the
// horizontal pass spent 24 of those per pixel and row)
__device__ __forceinline__ void mad24(int& acc, int px, int k) { asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(acc) : "v"(px), "v"(k)); }
// the same with a wave-uniform coefficient (an SGPR operand: no v_mov per tap)
__device__ __forceinline__ void mad24s(int& acc, int px, int k) { asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(acc) : "v"(px), "s"(k)); }

// grid = (ceil(h / RB), B), NTH threads.  LDS: srow[RB][wcp] | orow[RB][ocp]  (row bytes rounded up to 4).
// A thread owns output pixel xx (and xx + NTH, ...): its first tap and its <= KS coefficients stay in registers
// for the RB rows of the block.  The allocation carries H_SLACK bytes behind orow for the over-read described below.
constexpr int H_SLACK = 128;
constexpr int NTH = 384;         // 336 output pixels -> one pixel per thread, 87 % of the lanes busy
template <int C, int KS>
__global__ __launch_bounds__(NTH) void clip_h_rows_kernel(const uint8_t* __restrict__ src, int h, int w, int left,
                                                          int ow, const int32_t* __restrict__ bounds,
                                                          const int32_t* __restrict__ kk, int ksize, int RB,
                                                          int src_dwords_ok, uint8_t* __restrict__ tmp) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_b[];
  const int wc = w * C, oc = ow * C, wcp = (wc + 3) & ~3, ocp = (oc + 3) & ~3;
  uint8_t* srow = lds_b;
  uint8_t* orow = srow + (size_t)RB * wcp;
  const int tid = threadIdx.x, b = blockIdx.y;
  const int y0 = blockIdx.x * RB, nrows = min(RB, h - y0);

  const uint8_t* sp = src + ((size_t)b * h + y0) * wc;
  if (src_dwords_ok) {          // rows start on 4-byte boundaries: coalesced dword loads, all rows of the block in flight
    const int nd = wc >> 2;
    for (int d = tid; d < nd; d += NTH) {
      for (int r0 = 0; r0 < nrows; r0 += 8) {
        uint32_t v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = reinterpret_cast<const uint32_t*>(sp + (size_t)min(r0 + i, nrows - 1) * wc)[d];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (r0 + i < nrows) reinterpret_cast<uint32_t*>(srow + (size_t)(r0 + i) * wcp)[d] = v[i];
      }
    }
  } else {
    for (int r = 0; r < nrows; ++r)
      for (int d = tid; d < wc; d += NTH) srow[(size_t)r * wcp + d] = sp[(size_t)r * wc + d];
  }
  __syncthreads();

  // The KS taps x C channels of a pixel are KS*C CONSECUTIVE bytes of the source row starting at xmin*C: they are read
  // as KS*C/4 + 1 dwords, shifted into place with v_alignbyte_b32 and consumed with a static byte extract + one
  // v_mad_i32_i24 each (byte reads made the LDS the bottleneck: 24 ds_read_u8 per pixel and row for RGB, 41 % of the
  // LDS cycles bank conflicts -- profiles/round2_clip_pmc.txt).  Taps past the pixel's count carry a zero coefficient;
  // the bytes they multiply may lie past the row (the next row, or the slack behind the last one).
  constexpr int NB = KS * C, NA = NB / 4, NW = NA + 1;
  for (int xx = tid; xx < ow; xx += NTH) {
    const int xo = left + xx;
    const int xmin = bounds[2 * xo], cnt = bounds[2 * xo + 1];
    int kreg[KS];
#pragma unroll
    for (int x = 0; x < KS; ++x) kreg[x] = (x < cnt) ? kk[(size_t)xo * ksize + min(x, ksize - 1)] : 0;
    const int o = xmin * C, sh = o & 3;
    const uint8_t* first = srow + (o & ~3);
    for (int r = 0; r < nrows; ++r) {
      const uint32_t* rw = reinterpret_cast<const uint32_t*>(first + (size_t)r * wcp);
      uint32_t wv[NW], a[NA];
#pragma unroll
      for (int i = 0; i < NW; ++i) wv[i] = rw[i];
#pragma unroll
      for (int i = 0; i < NA; ++i) a[i] = __builtin_amdgcn_alignbyte(wv[i + 1], wv[i], sh);
      int ss[C];
#pragma unroll
      for (int c = 0; c < C; ++c) ss[c] = 1 << (PIL_PRECISION_BITS - 1);
#pragma unroll
      for (int x = 0; x < KS; ++x)
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const int j = x * C + c;
          mad24(ss[c], (int)((a[j >> 2] >> (8 * (j & 3))) & 0xffu), kreg[x]);
        }
      uint8_t* op = orow + (size_t)r * ocp + xx * C;
#pragma unroll
      for (int c = 0; c < C; ++c) op[c] = clip8(ss[c]);
    }
  }
  __syncthreads();

  uint8_t* tp = tmp + ((size_t)b * h + y0) * oc;
  if ((oc & 3) == 0) {          // tmp rows start on 4-byte boundaries (the caller's buffer is 4-byte aligned)
    const int nd = oc >> 2;
    for (int r = 0; r < nrows; ++r)
      for (int d = tid; d < nd; d += NTH)
        reinterpret_cast<uint32_t*>(tp + (size_t)r * oc)[d] = reinterpret_cast<const uint32_t*>(orow + (size_t)r * ocp)[d];
  } else {
    for (int r = 0; r < nrows; ++r)
      for (int d = tid; d < oc; d += NTH) tp[(size_t)r * oc + d] = orow[(size_t)r * ocp + d];
  }
}

// grid = (ceil(oh / RV), B); ow*C % 4 == 0, ow*C <= 4*NT*NQ.  LDS: lut[C*256] floats | tile[C*ow] OutT.
// All index arithmetic (which pixel / channel a lane's 4 bytes belong to, where its share of the planar row goes)
// is done once per workgroup, outside the row loop.
constexpr int NQ = 2;            // dwords of a tmp row per thread (ow*C <= 2048)
constexpr int NS = 8;            // output elements per thread in the store phase
template <typename OutT, int C>
__global__ __launch_bounds__(NT) void clip_v_rows_kernel(const uint8_t* __restrict__ tmp, int h, int top, int oh,
                                                         int ow, const int32_t* __restrict__ bounds,
                                                         const int32_t* __restrict__ kk, int ksize, float m0, float m1,
                                                         float m2, float m3, float s0, float s1, float s2, float s3,
                                                         OutT* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  float* lut = lds_f;
  OutT* tile = reinterpret_cast<OutT*>(lut + C * 256);
  const int tid = threadIdx.x, b = blockIdx.y;
  const int oc = ow * C, nq = oc >> 2;
  {
    const float mean[4] = {m0, m1, m2, m3}, stdv[4] = {s0, s1, s2, s3};
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float x = (float)((double)tid * (1.0 / 255.0));        // rescale: float32(float64(u8) * (1/255))
      lut[c * 256 + tid] = fsub(x, mean[c]) / stdv[c];             // normalize in float32
    }
  }
  int toff[NQ][4], lbase[NQ][4];           // where byte j of dword q lands in the planar tile / its channel's table
#pragma unroll
  for (int n = 0; n < NQ; ++n)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = 4 * (tid + NT * n) + j;
      const int xx = e / C, c = e - xx * C;
      toff[n][j] = c * ow + xx;
      lbase[n][j] = c * 256;
    }
  // 32-bit offsets inside one image (h * oc and C * oh * ow stay below 2^31 for every shape the entry point admits):
  // with 64-bit element indices every tap load and every store paid a 64-bit multiply-add
  const uint8_t* timg = tmp + (size_t)b * h * oc;
  OutT* oimg = out + (size_t)b * C * oh * ow;
  int ooff[NS];                            // planar offset of tile element tid + NT*n (without the row term)
#pragma unroll
  for (int n = 0; n < NS; ++n) {
    const int i = tid + NT * n;
    const int c = i / ow, xx = i - c * ow;
    ooff[n] = c * oh * ow + xx;
  }
  __syncthreads();
  const int yy0 = blockIdx.x * RV, yy1 = min(yy0 + RV, oh);
  for (int yy = yy0; yy < yy1; ++yy) {
    const int yo = top + yy;
    const int ymin = bounds[2 * yo], cnt = bounds[2 * yo + 1];
    const int32_t* k = kk + (size_t)yo * ksize;
    const uint8_t* col = timg + (unsigned)(ymin * oc);
    const int hlast = h - 1;
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      const int q = tid + NT * n;
      if (q < nq) {
        int a0 = 1 << (PIL_PRECISION_BITS - 1), a1 = a0, a2 = a0, a3 = a0;
        if (ksize == 8) {
          // rows of exactly 8 coefficients: one 32-byte scalar load, masked by the row's tap count (the caller's table
          // need not be zero-padded), and 8 row loads without a branch (row indices clamped to the image)
          uint32_t wv[8];
          int kv[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) kv[i] = i < cnt ? __builtin_amdgcn_readfirstlane(k[i]) : 0;
#pragma unroll
          for (int i = 0; i < 8; ++i) wv[i] = reinterpret_cast<const uint32_t*>(col + (unsigned)(min(i, hlast - ymin) * oc))[q];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            mad24s(a0, (int)(wv[i] & 0xffu), kv[i]);
            mad24s(a1, (int)((wv[i] >> 8) & 0xffu), kv[i]);
            mad24s(a2, (int)((wv[i] >> 16) & 0xffu), kv[i]);
            mad24s(a3, (int)(wv[i] >> 24), kv[i]);
          }
        } else
        for (int y0 = 0; y0 < cnt; y0 += 8) {
          // 8 tap rows requested before the first multiply (taps past the count re-read the last row with a zero
          // coefficient); one dependent round trip per 8 taps instead of per tap
          uint32_t wv[8];
          int kv[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int y = min(y0 + i, cnt - 1);
            wv[i] = reinterpret_cast<const uint32_t*>(col + (unsigned)(y * oc))[q];
            kv[i] = __builtin_amdgcn_readfirstlane((y0 + i < cnt) ? k[y] : 0);      // wave uniform
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            mad24s(a0, (int)(wv[i] & 0xffu), kv[i]);
            mad24s(a1, (int)((wv[i] >> 8) & 0xffu), kv[i]);
            mad24s(a2, (int)((wv[i] >> 16) & 0xffu), kv[i]);
            mad24s(a3, (int)(wv[i] >> 24), kv[i]);
          }
        }
        tile[toff[n][0]] = from_f32<OutT>(lut[lbase[n][0] + clip8(a0)]);
        tile[toff[n][1]] = from_f32<OutT>(lut[lbase[n][1] + clip8(a1)]);
        tile[toff[n][2]] = from_f32<OutT>(lut[lbase[n][2] + clip8(a2)]);
        tile[toff[n][3]] = from_f32<OutT>(lut[lbase[n][3] + clip8(a3)]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NS; ++n) {
      const int i = tid + NT * n;
      if (i < oc) oimg[ooff[n] + yy * ow] = tile[i];
    }
    __syncthreads();
  }
}

}  // namespace clip
}  // namespace attwarp

using namespace attwarp;

extern "C" int attwarp_clip_preprocess_u8(const uint8_t* src, int B, int h, int w, int C, int top, int left, int size,
                                          const int32_t* bounds_x, const int32_t* kk_x, int ksize_x,
                                          const int32_t* bounds_y, const int32_t* kk_y, int ksize_y,
                                          const float* mean, const float* stdv /* host, C floats each */,
                                          uint8_t* tmp, void* out, int out_dtype, void* stream) {
  ATTWARP_REQUIRE(src && bounds_x && kk_x && bounds_y && kk_y && mean && stdv && tmp && out,
                  "clip_preprocess_u8: null pointer");
  ATTWARP_REQUIRE(B > 0 && h > 0 && w > 0 && size > 0 && top >= 0 && left >= 0 && ksize_x > 0 && ksize_y > 0,
                  "clip_preprocess_u8: bad size");
  ATTWARP_REQUIRE(C >= 1 && C <= 4, "clip_preprocess_u8: C must be 1..4 (got %d)", C);
  ATTWARP_REQUIRE(out_dtype == ATTWARP_F32 || out_dtype == ATTWARP_F16, "clip_preprocess_u8: out dtype must be F32 or F16");
  if (B > 65535 || h > 65535 || size > 65535) return fail(ATTWARP_E_UNSUPPORTED, "clip_preprocess_u8: dims > 65535");
  hipStream_t st = as_stream(stream);
  float m[4] = {0, 0, 0, 0}, s[4] = {1, 1, 1, 1};
  for (int c = 0; c < C; ++c) { m[c] = mean[c]; s[c] = stdv[c]; }

  const bool force_generic = tune(TUNE_CLIP_VARIANT) == 1;
  const long long wc = (long long)w * C, oc = (long long)size * C;
  const long long wcp = (wc + 3) & ~3LL, ocp = (oc + 3) & ~3LL;
  long long RB = (48 * 1024) / (wcp + ocp);               // LDS per workgroup of the horizontal pass
  if (RB > 8) RB = 8;
  const size_t lds_v = (size_t)C * 256 * sizeof(float) + (size_t)oc * (out_dtype == ATTWARP_F32 ? 4 : 2);
  if (force_generic || C == 2 || RB < 1 || ksize_x > 16 || (oc & 3) != 0 || oc > 4 * clip::NT * clip::NQ ||
      oc > clip::NT * clip::NS || lds_v > 48 * 1024 || (reinterpret_cast<uintptr_t>(tmp) & 3u) != 0)
    return clip_preprocess_generic(src, B, h, w, C, top, left, size, bounds_x, kk_x, ksize_x, bounds_y, kk_y, ksize_y, m,
                                   s, tmp, out, out_dtype, st);

  const int src_dwords_ok = (wc % 4 == 0) && ((reinterpret_cast<uintptr_t>(src) & 3u) == 0);
  const size_t lds_h = (size_t)RB * (size_t)(wcp + ocp) + clip::H_SLACK;
  const dim3 gh((unsigned)((h + RB - 1) / RB), B), th(clip::NTH);
#define ATTWARP_CLIP_H(CC, KS)                                                                                     \
  hipLaunchKernelGGL((clip::clip_h_rows_kernel<CC, KS>), gh, th, lds_h, st, src, h, w, left, size, bounds_x, kk_x,    \
                     ksize_x, (int)RB, src_dwords_ok, tmp)
  if (ksize_x <= 8) {
    if (C == 1) ATTWARP_CLIP_H(1, 8); else if (C == 3) ATTWARP_CLIP_H(3, 8); else ATTWARP_CLIP_H(4, 8);
  } else {
    if (C == 1) ATTWARP_CLIP_H(1, 16); else if (C == 3) ATTWARP_CLIP_H(3, 16); else ATTWARP_CLIP_H(4, 16);
  }
#undef ATTWARP_CLIP_H
  int rc = check_launch("clip_h_rows_kernel");
  if (rc) return rc;
  const dim3 grid((size + clip::RV - 1) / clip::RV, B);
#define ATTWARP_CLIP_V(OT, CC)                                                                                     \
  hipLaunchKernelGGL((clip::clip_v_rows_kernel<OT, CC>), grid, dim3(clip::NT), lds_v, st, tmp, h, top, size, size,   \
                     bounds_y, kk_y, ksize_y, m[0], m[1], m[2], m[3], s[0], s[1], s[2], s[3], (OT*)out)
  if (out_dtype == ATTWARP_F32) {
    if (C == 1) ATTWARP_CLIP_V(float, 1); else if (C == 3) ATTWARP_CLIP_V(float, 3); else ATTWARP_CLIP_V(float, 4);
  } else {
    if (C == 1) ATTWARP_CLIP_V(__half, 1); else if (C == 3) ATTWARP_CLIP_V(__half, 3); else ATTWARP_CLIP_V(__half, 4);
  }
#undef ATTWARP_CLIP_V
  return check_launch("clip_v_rows_kernel");
}
