// K7 for uint8 images: the resample of the main_batched chain (AGW/new_method.py:268-271 on uint8 BGR
// images, typically 336x336 -> 500x500).  Same decomposition as remap_rows_kernel (remap_rows.hip): one
// workgroup owns R consecutive output rows of one image, vertical lerp first into an LDS float row, then
// the horizontal gather.  Arithmetic is the uint8 "exact" path of remap_gather_kernel / the oracle:
// bytes -> float32, three individually rounded lerps, round half to even (a lerp of values in [0,255] with a
// weight in [0,1) stays in [0,255] after every rounding, so no clamp is needed).
//
// CV2 mode (OpenCV's uint8 path: 1/32-pixel coordinates, int16 weights w = (32-ky|ky)*(32-kx|kx)*32 scaled to 2^15,
// result (sum + 2^14) >> 15) runs on the SAME structure: the weighted sum is an integer identity,
//   sum = 32 * [ (32-kx)*v0 + kx*v1 ],   v = (32-ky)*top + ky*bottom      (v <= 8160, [..] <= 255*1024)
//   (sum + 2^14) >> 15 = floor(([..] + 512) / 1024)
// and every intermediate is an integer below 2^24, i.e. exact in float32 whatever the association: the vertical
// pass stages v, the gather forms t = [..] + 0.5 and takes the low byte of fma(t, 2^-10, 2^23) (round to nearest
// of y + 2^-11 where y = [..]/1024 has 10 fractional bits: never a tie, equals floor(y + 1/2)).  OpenCV's
// saturation of the (ky,kx) = (0,0) weight 32768 -> 32767 is not observable ((p*32767 + 2^14) >> 15 = p).
//
// This kernel is bound by instruction issue, not by HBM (1500 output bytes per row, each needing two LDS taps
// and a lerp), so the work per byte is kept minimal and spread over all four waves:
//   * vertical pass: every thread owns dwords of the source row (4 bytes: global_load_dword, coalesced); both
//     source rows of the NEXT output row are fetched while the current row is gathered (static double
//     buffering; the re-read of a shared row is an L1/L2 hit, HBM sees every byte once); 4 lerps ->
//     one ds_write_b128 into the float row;
//   * gather: one output byte per lane per k-slice (consecutive lanes -> consecutive LDS banks), the byte goes
//     to an LDS output row; the row is written to global as dwords, 256 contiguous bytes per wave
//     instruction, after the NEXT row's barrier (the barrier between the two passes is the only one per row).
#include "common.hpp"
#include "remap_u8_block.hpp"

namespace attwarp {

namespace u8k {


struct Taps {
  int i0, i1;
  float f;      // EXACT: fractional part; CV2: k = q & 31 as a float (0 .. 31)
};
template <int MODE>
__device__ __forceinline__ Taps taps(float m, int size) {
  Taps t;
  int i;
  if (MODE == ATTWARP_CV2) {
    const int q = cv_round_q5(m);         // cvRound
    i = q >> 5;
    t.f = (float)(q & 31);
  } else {
    const float mc = clamp_coord(m, size);
    const float fl = floorf(mc);
    t.f = fsub(mc, fl);
    i = (int)fl;
  }
  t.i0 = min(max(i, 0), size - 1);
  t.i1 = min(max(i + 1, 0), size - 1);
  return t;
}
// vertical stage: EXACT a + f*(c-a) (three roundings); CV2 (32-k)*a + k*c (exact integers)
template <int MODE>
__device__ __forceinline__ float vblend(float a, float c, float f) {
  if (MODE == ATTWARP_CV2) return __fmaf_rn(f, fsub(c, a), fmul(32.0f, a));
  return lerp_rn(a, c, f);
}
// horizontal stage + rounding, result in the low byte of the returned float's bit pattern
template <int MODE>
__device__ __forceinline__ float hblend_biased(float v0, float v1, float f) {
  if (MODE == ATTWARP_CV2) {
    const float t = __fmaf_rn(f, fsub(v1, v0), __fmaf_rn(32.0f, v0, 0.5f));   // (32-k)*v0 + k*v1 + 1/2, exact
    return __fmaf_rn(t, 0.0009765625f, 8388608.0f);
  }
  return fadd(lerp_rn(v0, v1, f), 8388608.0f);   // round half to even: the low byte of (x + 2^23) for 0 <= x <= 255
}


// KI = dwords of the source row per thread, KO = output bytes per thread, KS = output dwords per thread
// TILED (rows wider than 4096 bytes, one plane): a workgroup owns a column tile of KO*NT output bytes of its rows,
// stages the source span [min tap, max tap] relative to its dword-aligned start; a span that does not fit KI*NT
// dwords is read tap by tap from global memory for that tile (same scheme as remap_rows_kernel's TILED).
template <int KI, int KO, bool HWC, bool TILED, int MODE>
__global__ __launch_bounds__(NT) void remap_rows_u8_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int KS = (KO + 3) / 4;
  float* s_my = smem;                      // RMAX
  float* row0 = smem + RMAX;               // 2 float rows of VLP floats
  const int VLP = TILED ? KI * NT * 4 : (p.VL + 15) & ~15;
  float* row1 = row0 + VLP;
  constexpr int OVP = NT * KO;                              // whole k-slices: no index clamp in the gather
  uint8_t* out0 = reinterpret_cast<uint8_t*>(row1 + VLP);   // 2 output rows of OVP bytes
  uint8_t* out1 = out0 + OVP;
  const int tid = threadIdx.x;

  int bid = blockIdx.x;
  {
    const int n = p.nblocks, q = n >> 3, r = n & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r) ? xcd * (q + 1) + idx : r * (q + 1) + (xcd - r) * q + idx;
  }
  int b, rb, tile = 0;
  if (TILED) {
    const int per_img = p.nblk * p.ntiles;
    b = bid / per_img;
    const int rem = bid - b * per_img;
    tile = rem / p.nblk;
    rb = rem - tile * p.nblk;
  } else {
    b = bid / p.nblk;
    rb = bid - b * p.nblk;
  }
  const int y0 = rb * p.R;
  const int nrows = min(y0 + p.R, p.Ho) - y0;
  const uint8_t* src_b = p.src + (long long)b * p.img_stride;
  uint8_t* dst_b = p.dst + (long long)b * p.oimg_stride;
  const int bm = b / p.map_div;

  if (tid < nrows) s_my[tid] = p.my[(long long)bm * p.Ho + y0 + tid];

  int goff[KI], voff[KI];   // byte offset inside the image (row 0) / float index in the LDS row of this thread's dwords
  unsigned pk[KO];
  float fxr[KO];
  int soff[KS], sld[KS];    // byte offset of this thread's output dwords inside an output row (incl. plane) / in the LDS row
  bool direct = false;
  unsigned f0s[TILED ? KO : 1], f1s[TILED ? KO : 1];
  if (TILED) {
    __shared__ int s_lo[NT / WAVE], s_hi[NT / WAVE];
    const int e0 = tile * (KO * NT), e1 = min(e0 + KO * NT, p.OVL);
    int lo = 0x7fffffff, hi = 0;
#pragma unroll
    for (int k = 0; k < KO; ++k) {
      const int e = min(e0 + tid + NT * k, e1 - 1);
      const int x = e / p.CS, c = e - x * p.CS;
      const Taps tx = taps<MODE>(p.mx[(long long)bm * p.Wo + x], p.W);
      f0s[k] = tx.i0 * p.CS + c;
      f1s[k] = tx.i1 * p.CS + c;
      fxr[k] = tx.f;
      lo = min(lo, (int)f0s[k]);
      hi = max(hi, (int)f1s[k]);
    }
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) {
      lo = min(lo, __shfl_xor(lo, o, WAVE));
      hi = max(hi, __shfl_xor(hi, o, WAVE));
    }
    if ((tid & (WAVE - 1)) == 0) { s_lo[tid / WAVE] = lo; s_hi[tid / WAVE] = hi; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NT / WAVE; ++w) { lo = min(lo, s_lo[w]); hi = max(hi, s_hi[w]); }
    const int abase = lo & ~3;                       // dword-aligned start of the staged span
    const int nd4 = (hi - abase + 4) >> 2;           // dwords covering [abase, hi]
    direct = nd4 > KI * NT;
#pragma unroll
    for (int k = 0; k < KO; ++k) pk[k] = ((f0s[k] - abase) * 4u) | (((f1s[k] - abase) * 4u) << 16);
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      const int d = min(tid + NT * k, nd4 - 1);
      goff[k] = abase + 4 * d;
      voff[k] = 4 * d;
    }
    const int nd = (e1 - e0) >> 2;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const int d = min(tid + NT * k, nd - 1);
      soff[k] = e0 + 4 * d;
      sld[k] = 4 * d;
    }
  } else {
  // source dwords this thread owns (clamped: padding lanes re-read the last dword and write the same floats)
  {
    const int dpr = p.row_len >> 2, nd = p.VL >> 2;
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      const int d = min(tid + NT * k, nd - 1);
      const int pl = HWC ? 0 : d / dpr;
      goff[k] = (int)(pl * p.plane_stride) + 4 * (d - pl * dpr);
      voff[k] = 4 * d;
    }
  }
  // column taps of the output bytes this thread gathers.  Interleaved rows: element e = x*CS + c advances by NT
  // per k-slice, so (x, c) is stepped with one carry instead of a division per element.
  {
    const int dx = NT / p.CS, dc = NT - dx * p.CS;       // block uniform
    int x = HWC ? tid / p.CS : 0, c = HWC ? tid - x * p.CS : 0;
#pragma unroll
    for (int k = 0; k < KO; ++k) {
      int pl = 0, xe = x, ce = c;
      if (HWC) {
        if (tid + NT * k > p.OVL - 1) { xe = p.Wo - 1; ce = p.CS - 1; }     // padding lanes repeat the last element
        x += dx; c += dc;
        if (c >= p.CS) { c -= p.CS; ++x; }
      } else {                                            // planar: CS == 1
        const int e = min(tid + NT * k, p.OVL - 1);
        pl = e / p.orow_len;
        xe = e - pl * p.orow_len;
        ce = 0;
      }
      const Taps tx = taps<MODE>(p.mx[(long long)bm * p.Wo + xe], p.W);
      const unsigned i0 = pl * p.row_len + tx.i0 * p.CS + ce;
      const unsigned i1 = pl * p.row_len + tx.i1 * p.CS + ce;
      pk[k] = (i0 * 4u) | ((i1 * 4u) << 16);
      fxr[k] = tx.f;
    }
  }
  // output dwords this thread stores (clamped: padding lanes repeat the last)
  {
    const int dpo = p.orow_len >> 2, nd = p.OVL >> 2;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const int d = min(tid + NT * k, nd - 1);
      const int pl = HWC ? 0 : d / dpo;
      soff[k] = (int)(pl * p.oplane_stride) + 4 * (d - pl * dpo);
      sld[k] = 4 * d;
    }
  }
  }
  __syncthreads();

  if (TILED && direct) {     // block uniform: this tile's source span does not fit the staged row
    const int e0 = tile * (KO * NT), e1 = min(e0 + KO * NT, p.OVL);
    for (int q = 0; q < nrows; ++q) {
      const Taps ty = taps<MODE>(s_my[q], p.H);
      const uint8_t* ra = src_b + (long long)ty.i0 * p.row_len;
      const uint8_t* rc = src_b + (long long)ty.i1 * p.row_len;
      uint8_t* orow = dst_b + (long long)(y0 + q) * p.orow_len;
#pragma unroll
      for (int k = 0; k < KO; ++k) {
        const int e = e0 + tid + NT * k;
        const float v0 = vblend<MODE>((float)ra[f0s[k]], (float)rc[f0s[k]], ty.f);
        const float v1 = vblend<MODE>((float)ra[f1s[k]], (float)rc[f1s[k]], ty.f);
        if (e < e1) orow[e] = (uint8_t)__float_as_uint(hblend_biased<MODE>(v0, v1, fxr[k]));
      }
    }
    return;
  }

  Taps tcur = taps<MODE>(s_my[0], p.H);
  uint32_t A[KI], C[KI];
#define ATTWARP_U8_FETCH()                                                                             \
  _Pragma("unroll") for (int k = 0; k < KI; ++k) {                                                      \
    A[k] = *reinterpret_cast<const uint32_t*>(src_b + (long long)tcur.i0 * p.row_len + goff[k]);        \
    C[k] = *reinterpret_cast<const uint32_t*>(src_b + (long long)tcur.i1 * p.row_len + goff[k]);        \
  }
  // the output row finished one iteration ago: LDS bytes -> global dwords
#define ATTWARP_U8_FLUSH(obuf, yrow)                                                                   \
  {                                                                                                    \
    uint8_t* orow_ = dst_b + (long long)(yrow) * p.orow_len;                                            \
    _Pragma("unroll") for (int k = 0; k < KS; ++k)                                                      \
        *reinterpret_cast<uint32_t*>(orow_ + soff[k]) = *reinterpret_cast<const uint32_t*>((obuf) + sld[k]);  \
  }
  // one output row: vertical lerp -> rowbuf, prefetch, barrier, flush the previous row, gather -> outbuf.
  // rowbuf / outbuf / prevbuf are the two fixed LDS buffers (immediate offsets), hence the 2x unrolled loop below.
#define ATTWARP_U8_ROW(q_, rowbuf, outbuf, prevbuf)                                                    \
  {                                                                                                    \
    const float fy_ = tcur.f;                                                                          \
    _Pragma("unroll") for (int k = 0; k < KI; ++k) {                                                    \
      float4 v_;                                                                                       \
      v_.x = vblend<MODE>((float)(A[k] & 0xffu), (float)(C[k] & 0xffu), fy_);                          \
      v_.y = vblend<MODE>((float)((A[k] >> 8) & 0xffu), (float)((C[k] >> 8) & 0xffu), fy_);            \
      v_.z = vblend<MODE>((float)((A[k] >> 16) & 0xffu), (float)((C[k] >> 16) & 0xffu), fy_);          \
      v_.w = vblend<MODE>((float)(A[k] >> 24), (float)(C[k] >> 24), fy_);                              \
      *reinterpret_cast<float4*>((rowbuf) + voff[k]) = v_;                                             \
    }                                                                                                  \
    if ((q_) + 1 < nrows) { /* fetch the next output row's two source rows now */                      \
      tcur = taps<MODE>(s_my[(q_) + 1], p.H);                                                                \
      ATTWARP_U8_FETCH()                                                                               \
    }                                                                                                  \
    __syncthreads();                                                                                   \
    if ((q_) > 0) ATTWARP_U8_FLUSH(prevbuf, y0 + (q_) - 1)                                             \
    const char* rowb_ = reinterpret_cast<const char*>(rowbuf);                                         \
    _Pragma("unroll") for (int k = 0; k < KO; ++k) {                                                    \
      unsigned w_ = pk[k];                                                                             \
      asm volatile("" : "+v"(w_));                                                                     \
      const float v0_ = *reinterpret_cast<const float*>(rowb_ + (w_ & 0xffffu));                       \
      const float v1_ = *reinterpret_cast<const float*>(rowb_ + (w_ >> 16));                           \
      const float r_ = hblend_biased<MODE>(v0_, v1_, fxr[k]);                                          \
      (outbuf)[tid + NT * k] = (uint8_t)__float_as_uint(r_);                                           \
    }                                                                                                  \
  }
  ATTWARP_U8_FETCH()
  int q = 0;
  for (; q + 1 < nrows; q += 2) {
    ATTWARP_U8_ROW(q, row0, out0, out1)
    ATTWARP_U8_ROW(q + 1, row1, out1, out0)
  }
  if (q < nrows) ATTWARP_U8_ROW(q, row0, out0, out1)
  __syncthreads();
  if ((nrows - 1) & 1) ATTWARP_U8_FLUSH(out1, y0 + nrows - 1) else ATTWARP_U8_FLUSH(out0, y0 + nrows - 1)
#undef ATTWARP_U8_ROW
#undef ATTWARP_U8_FLUSH
#undef ATTWARP_U8_FETCH
}

template <int KI, int KO, int MODE>
static int launch_kiko(const Params& p, hipStream_t st) {
  const int VLP = (p.VL + 15) & ~15, OVP = NT * KO;
  const size_t lds = (size_t)(RMAX + 2 * VLP) * sizeof(float) + 2 * (size_t)OVP;
  if (p.NP == 1)
    hipLaunchKernelGGL((remap_rows_u8_kernel<KI, KO, true, false, MODE>), dim3(p.nblocks), dim3(NT), lds, st, p);
  else
    hipLaunchKernelGGL((remap_rows_u8_kernel<KI, KO, false, false, MODE>), dim3(p.nblocks), dim3(NT), lds, st, p);
  return check_launch("remap_rows_u8_kernel");
}

template <int KO, int MODE>
static int launch_ko(const Params& p, hipStream_t st) {
  const int ki = ((p.VL >> 2) + NT - 1) / NT;
  switch (ki) {
    case 1: return launch_kiko<1, KO, MODE>(p, st);
    case 2: return launch_kiko<2, KO, MODE>(p, st);
    case 3: return launch_kiko<3, KO, MODE>(p, st);
    default: return launch_kiko<4, KO, MODE>(p, st);
  }
}

// ---- CV2 mode, integer form (rows of <= 4096 bytes): body remap_rows_u8i_block (remap_u8_block.hpp) ----
template <int KI, int KD, bool HWC, int PD, bool UA>
__global__ __launch_bounds__(NT) void remap_rows_u8i_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  remap_rows_u8i_block<KI, KD, HWC, PD, UA>(p, blockIdx.x, smem);
}

template <int KI, int PD>
static int launch_u8i_ki(const Params& p, hipStream_t st) {
  const size_t lds = (size_t)RMAX * sizeof(float) + 2 * (size_t)U8I_VLP * sizeof(uint16_t);
  const int kd = ((((p.OVL + 3) >> 2)) + NT - 1) / NT;
  const dim3 g(p.nblocks), t(NT);
#define ATTWARP_U8I_LAUNCH(KD)                                                                              \
  if (p.unaligned) hipLaunchKernelGGL((remap_rows_u8i_kernel<KI, KD, true, PD, true>), g, t, lds, st, p);    \
  else if (p.NP == 1) hipLaunchKernelGGL((remap_rows_u8i_kernel<KI, KD, true, PD, false>), g, t, lds, st, p); \
  else hipLaunchKernelGGL((remap_rows_u8i_kernel<KI, KD, false, PD, false>), g, t, lds, st, p)
  if (kd <= 1) { ATTWARP_U8I_LAUNCH(1); } else if (kd == 2) { ATTWARP_U8I_LAUNCH(2); }
  else if (kd == 3) { ATTWARP_U8I_LAUNCH(3); } else { ATTWARP_U8I_LAUNCH(4); }
#undef ATTWARP_U8I_LAUNCH
  return check_launch("remap_rows_u8i_kernel");
}
// Rows requested ahead.  Measured on MI355X, one process cycling the variants (tools/attic/u8_ahead.py): rows of <= 2 KB
// (336x3: 1 KB) want 4 -- B=256 336 -> 500: 113.6 us with 1 row ahead, 103.9 with 2, 94.6 with 4; planar 336 -> 336:
// 53.5 / 51.2 / 48.9 -- rows of 3 KB (1024x3) 2: 1024 -> 500 161.8 / 153.8 / 153.0 at B=256 but 42.6 / 41.5 / 44.4 at B=64;
// 1024 -> 1024 is the same with all three (it follows the lease: 308 - 357 us).
template <int KI>
static int launch_u8i_depth(const Params& p, hipStream_t st) {
#ifdef ATTWARP_TUNING
  if (tune(TUNE_U8_AHEAD) == 1) return launch_u8i_ki<KI, 1>(p, st);
  if (tune(TUNE_U8_AHEAD) == 2) return launch_u8i_ki<KI, 2>(p, st);
  if (tune(TUNE_U8_AHEAD) == 4) return launch_u8i_ki<KI, 4>(p, st);
#endif
  return launch_u8i_ki<KI, (KI <= 2 ? 4 : 2)>(p, st);
}
static int launch_u8i(const Params& p, hipStream_t st) {
  const int ki = (((p.VL + 3) >> 2) + NT - 1) / NT;
  switch (ki) {
    case 1: return launch_u8i_depth<1>(p, st);
    case 2: return launch_u8i_depth<2>(p, st);
    case 3: return launch_u8i_depth<3>(p, st);
    default: return launch_u8i_depth<4>(p, st);
  }
}

template <int MODE>
static int launch_mode(const Params& p, bool tiled, hipStream_t st) {
  constexpr int TILE_KO = 8, TILE_KI = 3;      // 2048 output bytes per tile against 3072 staged source bytes
  if (tiled) {
    const size_t lds = (size_t)(RMAX + 2 * TILE_KI * NT * 4) * sizeof(float) + 2 * (size_t)(NT * TILE_KO);
    hipLaunchKernelGGL((remap_rows_u8_kernel<TILE_KI, TILE_KO, true, true, MODE>), dim3(p.nblocks), dim3(NT), lds, st, p);
    return check_launch("remap_rows_u8_kernel");
  }
  const int ko = (p.OVL + NT - 1) / NT;
  if (ko <= 4) return launch_ko<4, MODE>(p, st);
  if (ko <= 8) return launch_ko<8, MODE>(p, st);
  if (ko <= 12) return launch_ko<12, MODE>(p, st);
  return launch_ko<16, MODE>(p, st);
}

}  // namespace u8k

// Launch geometry of the uint8 staged kernels.  Returns false when the request takes the generic gather kernel;
// otherwise fills p and says which family serves it (*tiled: column tiles of the float pipeline; *integer_form: the
// integer cv2 kernel).
static bool plan_u8(u8k::Params& p, const uint8_t* src, uint8_t* dst, int layout, int B, int C, int H, int W, int Ho, int Wo,
                    const float* mx, const float* my, int mode, bool* tiled_out, bool* integer_out) {
  if (tune(TUNE_REMAP_VARIANT) == 1) return false;
  p.src = src; p.dst = dst; p.mx = mx; p.my = my;
  p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo;
  if (layout == ATTWARP_HWC) { p.NP = 1; p.CS = C; } else { p.NP = C; p.CS = 1; }
  p.row_len = W * p.CS;
  p.orow_len = Wo * p.CS;
  long long VL = (long long)p.NP * p.row_len, OVL = (long long)p.NP * p.orow_len;
  // dword loads / stores: every plane row starts on a 4-byte boundary on both sides -- or the request is "unaligned"
  // (e.g. 683 pixels x 3 bytes per row): served by the UA form of the integer cv2 kernel when its other conditions hold
  // (cv2 mode, rows of 4 .. 4096 bytes; planar images plane by plane), by the generic gather kernel otherwise (exact
  // mode, column-tiled rows)
  bool ua = p.row_len % 4 != 0 || p.orow_len % 4 != 0 ||
            ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 3u) != 0 ||
            (layout == ATTWARP_HWC ? ((long long)H * W * C) % 4 != 0 || ((long long)Ho * Wo * C) % 4 != 0
                                   : ((long long)H * W) % 4 != 0 || ((long long)Ho * Wo) % 4 != 0);
  if (tune(TUNE_BOUND) > 0 && (tune(TUNE_BOUND) & 16) && mode == ATTWARP_CV2 && layout == ATTWARP_HWC) ua = true;   // experiment: the UA form on aligned data
  if (ua && (mode != ATTWARP_CV2 || tune(TUNE_REMAP_VARIANT) == 2)) return false;
  p.map_div = 1;
  p.ntiles = 1;
  p.grp = tune(TUNE_REMAP_NOSWZ) >= 0 ? tune(TUNE_REMAP_NOSWZ) : 0;
  // up to 4096 bytes per staged row (16-bit LDS offsets); wider rows run in column tiles, planar ones plane by plane
  bool tiled = VL > 4096 || OVL > 4096;
  if (ua && layout != ATTWARP_HWC && p.row_len <= 4096 && p.orow_len <= 4096) tiled = false;   // plane by plane below
  if (ua && (tiled || p.row_len < 4)) return false;
  if (tiled || (ua && layout != ATTWARP_HWC)) {
    if (tiled && tune(TUNE_REMAP_TILED) == 0) return false;
    if ((long long)p.row_len > 2147483647LL / 8 || (long long)p.orow_len > 2147483647LL / 8 ||
        (long long)B * C > 2147483647LL)
      return false;
    if (p.NP > 1) {          // every plane becomes a one-channel image served by the maps of image b / C
      p.map_div = C;
      B *= C;
      C = 1;
      p.NP = 1;
      p.CS = 1;
      layout = ATTWARP_HWC;
      VL = p.row_len; OVL = p.orow_len;
    }
  }
  p.unaligned = ua ? 1 : 0;
  p.VL = (int)(tiled ? p.row_len : VL);
  p.OVL = (int)(tiled ? p.orow_len : OVL);
  p.plane_stride = (layout == ATTWARP_HWC) ? 0 : (long long)H * W;
  p.oplane_stride = (layout == ATTWARP_HWC) ? 0 : (long long)Ho * Wo;
  p.img_stride = (long long)H * W * C;
  p.oimg_stride = (long long)Ho * Wo * C;
  if (p.plane_stride * p.NP > 2147483647LL || p.oplane_stride * p.NP > 2147483647LL) return false;
  int R = (OVL >= 2048) ? 32 : 16;   // measured: 1024x1024x3 R=32 7 % faster than 16, 336->500 equal
  constexpr int TILE_KO = 8;                   // as in launch_mode
  // (2048x2048x3 uint8, B=64: tiled R=32 0.51 ms, R=16 0.55, R=8 0.67; generic gather kernel 2.32 ms)
  if (tiled) { R = 32; p.ntiles = (p.OVL + TILE_KO * u8k::NT - 1) / (TILE_KO * u8k::NT); }
  if (const int v = tune(TUNE_REMAP_ROWS); v >= 1 && v <= u8k::RMAX) R = v;
  if (R > Ho) R = Ho;
  p.R = R;
  p.nblk = (Ho + R - 1) / R;
  p.wpi = p.nblk;
  // (the integer kernel addresses an image through 32-bit buffer offsets)
  const bool integer_form = mode == ATTWARP_CV2 && !tiled && tune(TUNE_REMAP_VARIANT) != 2 &&
                            p.img_stride <= 2147483647LL && p.oimg_stride <= 2147483647LL;
  if (integer_form) {
    // Large rows that are not strongly minified: smaller row blocks, several per workgroup (the column-tap prologue is
    // still paid once per 64 rows, the rows in flight form a compact window).  Measured on MI355X, B=256 1024 -> 1024
    // x3 uint8 (tools/attic/u8_sweep.py): near-identity maps R=32 x 1: 425 us, R=16 x 4: 370; peaked maps 383 -> 358;
    // 1024 -> 500 and 336 -> 500 are fastest as they were (R=32 / 16, one block per workgroup).
    int cpw = 1;                                      // row blocks per workgroup (strided inside the image)
    if (OVL >= 2048 && 2LL * H <= 3LL * Ho && tune(TUNE_REMAP_ROWS) < 1) {
      p.R = R = Ho < 16 ? Ho : 16;
      p.nblk = (Ho + R - 1) / R;
      cpw = 4;
    }
    if (const int v = tune(TUNE_REMAP_CPW); v >= 1) cpw = v;
    while (cpw > 1 && (long long)((p.nblk + cpw - 1) / cpw) * B < 4096) cpw >>= 1;       // keep the chip filled
    p.wpi = (p.nblk + cpw - 1) / cpw;
  }
  if (ua && !integer_form) return false;
  const long long nb = (long long)p.wpi * B * p.ntiles;
  if (nb > 2147483647LL) return false;
  p.nblocks = (int)nb;
  *tiled_out = tiled;
  *integer_out = integer_form;
  return true;
}

// Returns via *handled whether the uint8 fast path took the request.
int launch_remap_rows_u8(const uint8_t* src, uint8_t* dst, int layout, int B, int C, int H, int W, int Ho, int Wo,
                         const float* mx, const float* my, int mode, hipStream_t st, bool* handled) {
  u8k::Params p;
  bool tiled = false, integer_form = false;
  *handled = plan_u8(p, src, dst, layout, B, C, H, W, Ho, Wo, mx, my, mode, &tiled, &integer_form);
  if (!*handled) return ATTWARP_OK;
  if (integer_form) return u8k::launch_u8i(p, st);
  if (mode == ATTWARP_CV2) return u8k::launch_mode<ATTWARP_CV2>(p, tiled, st);
  return u8k::launch_mode<ATTWARP_EXACT>(p, tiled, st);
}

bool u8i_params(u8k::Params& p, const uint8_t* src, uint8_t* dst, int layout, int B, int C, int H, int W, int Ho, int Wo,
                const float* mx, const float* my) {
  bool tiled = false, integer_form = false;
  return plan_u8(p, src, dst, layout, B, C, H, W, Ho, Wo, mx, my, ATTWARP_CV2, &tiled, &integer_form) && integer_form;
}

}  // namespace attwarp
