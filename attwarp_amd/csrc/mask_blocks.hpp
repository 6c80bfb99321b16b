// A3 (revise_mask, AGW/attention_extraction/llava.py:207-238) and A4 (ToPILImage x255 + PIL LANCZOS up-sampling of the
// 24 x 24 mask, llava.py:192-196,243,253) as device BLOCKS: bodies shared by the stand-alone kernels of attn.hip and the
// one-launch step of the main_batched chain (chain_step.hip), which runs them as block ranges of one grid.  LDS comes
// from the caller so that the chain kernel can give every body the same pool.
#pragma once
#include "common.hpp"

namespace attwarp {

constexpr int MASK_NT = 256;

// ---- A3: one workgroup per mask -----------------------------------------------------------
// LDS (caller provided): x 32*32 floats, red MASK_NT/64 doubles, fred 2*MASK_NT/64 floats -- mask_postproc_lds_bytes()
constexpr size_t mask_postproc_lds_bytes() { return 32 * 32 * sizeof(float) + (MASK_NT / WAVE) * sizeof(double) + 2 * (MASK_NT / WAVE) * sizeof(float); }
__device__ __forceinline__ void mask_postproc_block(const float* __restrict__ mask, int n, int ks, float coe,
                                                    float* __restrict__ out, int b, float* x, double* red, float* fred_) {
  constexpr int NT = MASK_NT;
  float (*fred)[NT / WAVE] = reinterpret_cast<float (*)[NT / WAVE]>(fred_);
  const int cnt = n * n;
  const float* m = mask + (size_t)b * cnt;
  float mn = INFINITY, mx = -INFINITY;
  for (int k = threadIdx.x; k < cnt; k += NT) {
    const float v = m[k];
    x[k] = v;
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
  }
  mn = wave_min(mn);
  mx = wave_max(mx);
  if ((threadIdx.x & (WAVE - 1)) == 0) {
    fred[0][threadIdx.x / WAVE] = mn;
    fred[1][threadIdx.x / WAVE] = mx;
  }
  __syncthreads();
  mn = fred[0][0]; mx = fred[1][0];
  for (int w = 1; w < NT / WAVE; ++w) { mn = fminf(mn, fred[0][w]); mx = fmaxf(mx, fred[1][w]); }
  // normalize("min"): (mat - min) / (max - min)
  const float range = fsub(mx, mn);
  double acc = 0.0;
  for (int k = threadIdx.x; k < cnt; k += NT) {
    const float v = fsub(x[k], mn) / range;
    x[k] = v;
    acc += (double)v;
  }
  // enhance: mat - mean ; / std (unbiased) ; * coe ; sigmoid ; clamp(0,1)
  const float mean = (float)(block_sum(acc, red) / (double)cnt);
  acc = 0.0;
  for (int k = threadIdx.x; k < cnt; k += NT) {
    const float v = fsub(x[k], mean);
    x[k] = v;
    acc += (double)v;
  }
  const double mu2 = block_sum(acc, red) / (double)cnt;
  acc = 0.0;
  for (int k = threadIdx.x; k < cnt; k += NT) {
    const double d = (double)x[k] - mu2;
    acc += d * d;
  }
  const float sd = (float)sqrt(block_sum(acc, red) / (double)(cnt - 1));
  for (int k = threadIdx.x; k < cnt; k += NT) {
    float v = x[k] / sd;
    v = fmul(v, coe);
    v = (float)(1.0 / (1.0 + exp(-(double)v)));
    x[k] = (v != v) ? v : fminf(fmaxf(v, 0.0f), 1.0f);     // torch.clamp propagates NaN (a constant map: 0 / 0 above)
  }
  __syncthreads();
  // Conv2d(1,1,ks,padding=(ks-1)/2,padding_mode="replicate"), all weights 1/ks^2
  const int pad = (ks - 1) / 2;
  const float wgt = 1.0f / (float)(ks * ks);
  for (int k = threadIdx.x; k < cnt; k += NT) {
    const int r = k / n, c = k - r * n;
    double a = 0.0;
    for (int dy = -pad; dy <= pad; ++dy)
      for (int dx = -pad; dx <= pad; ++dx) {
        const int rr = min(max(r + dy, 0), n - 1), cc = min(max(c + dx, 0), n - 1);
        a += (double)fmul(x[rr * n + cc], wgt);
      }
    out[(size_t)b * cnt + k] = (float)a;
  }
}

// ---- A4: Pillow's 8-bit separable resampler (ImagingResampleHorizontal/Vertical_8bpc) ----
constexpr int PIL_PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ uint8_t pil_clip8(int v) { return (uint8_t)min(max(v >> PIL_PRECISION_BITS, 0), 255); }
// four clip8 results packed into a dword: gfx950's v_ashr_pk_u8_i32 shifts, saturates to [0,255] and packs two
// values into the low 16 bits.  Used through the builtin with an explicit 16-bit mask: when the compiler forms the
// instruction by itself from `clip8(a) | clip8(b) << 8 | clip8(c) << 16 | ...` it omits that mask (ROCm 7.2) and the
// stale upper half of the destination register corrupts bytes 2 and 3.
__device__ __forceinline__ uint32_t pil_clip8x4(int s0, int s1, int s2, int s3) {
  const uint32_t lo = (uint32_t)(uint16_t)__builtin_amdgcn_ashr_pk_u8_i32(s0, s1, PIL_PRECISION_BITS);
  const uint32_t hi = (uint32_t)(uint16_t)__builtin_amdgcn_ashr_pk_u8_i32(s2, s3, PIL_PRECISION_BITS);
  return lo | (hi << 16);
}

// ToPILImage on a float tensor: pic.mul(255).byte()  (truncating cast)
__device__ __forceinline__ uint8_t to_pil_u8(float v) {
  const float s = fmul(v, 255.0f);
  return (uint8_t)min(max((int)truncf(s), 0), 255);
}

// Up-sampling form of the fused kernel (<= 8 taps on both axes: the 24 x 24 token grid blown up to the image size).
// The cost is the vertical pass -- out_h * out_w outputs of <= 8 taps each -- so the decomposition is chosen for it:
//   * a workgroup owns a COLUMN STRIP of 256 pixels (64 dwords, one per lane) and a chunk of the output rows; its
//     horizontal pass therefore produces each tile value it needs exactly once (h x 256 bytes of LDS) -- in the
//     row-block form every block of R rows recomputed ~8 of the 24 tile rows at full width, a third of all VALU work
//     (SQ_INSTS_VALU, profiles/round2_chain_pmc.txt);
//   * the four waves split the chunk's rows; everything that depends on the output row only (bounds, the 8
//     zero-padded coefficients) is wave uniform and comes from scalar loads issued one row ahead; a lane keeps the 8
//     tile rows of the current tap window unpacked in registers (the window slides by one tile row every
//     out_h / h rows), so an output dword costs 28-32 v_mad_i32_i24 (8-bit pixels, coefficients < 2^23), two
//     v_ashr_pk_u8_i32 and one store.
// Same integer arithmetic as the other forms (bit-identical to Pillow).  grid = (nstrips * nchunks, B).
// LDS: src[h*w] | tile[h][256].   kk_y rows hold exactly 8 coefficients (zero padded).
struct LanczosStripArgs {
  const float* mf;           // float mask in [0,1] (quantised x255 on the fly) or
  const uint8_t* mu;         // uint8 mask
  int h, w, out_h, out_w;
  const int32_t* bounds_x; const int32_t* kk_x; int ksize_x;
  const int32_t* bounds_y; const int32_t* kk_y;
  int nchunks, rows_per_chunk;
  uint8_t* out;
};
inline size_t lanczos_strip_lds_bytes(int h, int w) { return (size_t)((h * w + 3) & ~3) + (size_t)h * MASK_NT; }
// bx = strip * nchunks + chunk; mask_off: element offset of this image's h x w mask in a.mf / a.mu; out_img: this image's
// [out_h,out_w] bytes (a.out is not read here); lz: lanczos_strip_lds_bytes(h, w) bytes of LDS.
// UA ("unaligned"): out_w % 4 != 0 or an image that starts anywhere -- the dword stores stay dwords relative to the row
// start (unaligned access mode), the row's last 1..3 bytes are stored byte by byte by the lane that owns them.
template <int KS, bool UA = false>
__device__ __forceinline__ void lanczos_strip_block(const LanczosStripArgs& a, int bx, size_t mask_off, uint8_t* out_img, uint8_t* lz) {
  constexpr int NT = MASK_NT;
  const float* __restrict__ mf = a.mf; const uint8_t* __restrict__ mu = a.mu;
  const int h = a.h, w = a.w, out_h = a.out_h, out_w = a.out_w, ksize_x = a.ksize_x, nchunks = a.nchunks,
            rows_per_chunk = a.rows_per_chunk;
  // the coefficient tables are read-only for the lifetime of the launch: the constant address space tells the compiler
  // so whatever the kernel's argument list looks like, and wave-uniform reads of them stay scalar loads (as generic
  // pointers inside an argument struct they may alias `out`: vector loads, 19 more VGPRs)
  typedef const __attribute__((address_space(4))) int32_t* ctab;
  const ctab bounds_x = (ctab)a.bounds_x, kk_x = (ctab)a.kk_x, bounds_y = (ctab)a.bounds_y, kk_y = (ctab)a.kk_y;
  constexpr int SW = NT;                               // strip width in pixels
  const int srcp = (h * w + 3) & ~3;
  uint8_t* src = lz;
  uint8_t* tile = lz + srcp;
  const int tid = threadIdx.x;
  const int strip = bx / nchunks, chunk = bx - strip * nchunks;
  const int yy0 = chunk * rows_per_chunk, yy1 = min(yy0 + rows_per_chunk, out_h);
  const int ys = bounds_y[2 * yy0];                                          // first / one-past-last source row
  const int ye = bounds_y[2 * (yy1 - 1)] + bounds_y[2 * (yy1 - 1) + 1];
  const size_t ib = mask_off;
  for (int i = ys * w + tid; i < ye * w; i += NT) src[i] = mf ? to_pil_u8(mf[ib + i]) : mu[ib + i];
  __syncthreads();
  {   // horizontal pass: one column per thread
    const int xx = min(strip * SW + tid, out_w - 1);
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
      tile[r * SW + tid] = pil_clip8(ss);
    }
  }
  __syncthreads();
  const int lane = tid & (WAVE - 1);
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nq = out_w >> 2;
  const int q = strip * (SW / 4) + lane;               // global dword column
  const int nrows = yy1 - yy0;
  const int rows_per_wave = (nrows + NT / WAVE - 1) / (NT / WAVE);
  const int r_beg = yy0 + wid * rows_per_wave, r_end = min(r_beg + rows_per_wave, yy1);
  if (r_beg >= r_end) return;
  int win[8][4];
  int base = -0x40000000;
#pragma unroll
  for (int y = 0; y < 8; ++y) win[y][0] = win[y][1] = win[y][2] = win[y][3] = 0;
  auto load_row = [&](int r, int (&dst)[4]) {
    const uint32_t wv = reinterpret_cast<const uint32_t*>(tile + min(r, h - 1) * SW)[lane];
    dst[0] = (int)(wv & 0xffu); dst[1] = (int)((wv >> 8) & 0xffu); dst[2] = (int)((wv >> 16) & 0xffu); dst[3] = (int)(wv >> 24);
  };
  // the NEXT row's bounds and coefficients are fetched while the current row is computed
  int kc[8], ymin_c = bounds_y[2 * r_beg];
#pragma unroll
  for (int y = 0; y < 8; ++y) kc[y] = kk_y[(size_t)r_beg * 8 + y];
  // (UA: a narrow last strip may have no whole dword at all: nq == 0)
  uint8_t* __restrict__ orow = out_img + (size_t)r_beg * out_w + 4 * (UA ? min(q, nq) : min(q, nq - 1));
  const int tail_bytes = (UA && q == nq) ? (out_w & 3) : 0;
  int round_half = 1 << (PIL_PRECISION_BITS - 1);
  asm volatile("" : "+v"(round_half));                      // one register for the whole loop (not an inline constant: VOP3 takes none)
  for (int yy = r_beg; yy < r_end; ++yy) {
    const int ymin = ymin_c;
    int kv[8];
#pragma unroll
    for (int y = 0; y < 8; ++y) kv[y] = kc[y];
    const int yn = min(yy + 1, out_h - 1);
    ymin_c = bounds_y[2 * yn];
#pragma unroll
    for (int y = 0; y < 8; ++y) kc[y] = kk_y[(size_t)yn * 8 + y];
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
    // one v_mad_i32_i24 per tap and pixel (left to itself the compiler pairs v_mul_i32_i24 with v_add3_u32: 1.5
    // instructions per tap); the coefficient is wave uniform and sits in an SGPR.  The first tap takes Pillow's rounding
    // constant from a loop-invariant register as its addend (as "sum = constant; sum += ..." every output row paid four
    // v_mov_b32 to re-materialise it: a tenth of this loop's instructions)
    int s0, s1, s2, s3;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(s0) : "v"(win[0][0]), "s"(kv[0]), "v"(round_half));
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(s1) : "v"(win[0][1]), "s"(kv[0]), "v"(round_half));
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(s2) : "v"(win[0][2]), "s"(kv[0]), "v"(round_half));
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(s3) : "v"(win[0][3]), "s"(kv[0]), "v"(round_half));
#pragma unroll
    for (int y = 1; y < 8; ++y) {
      // Up-sampling never has more than 6 non-zero taps (support 3: int(c + 3.5) - int(c - 2.5) = 6 source pixels; ksize = 7
      // is Pillow's allocation bound): the zero 7th and the padded 8th are skipped (wave uniform; adding 0 * pixel is exact)
      if (y == 6 && kv[6] == 0 && kv[7] == 0) break;
      if (y == 7 && kv[7] == 0) break;
      asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(s0) : "v"(win[y][0]), "s"(kv[y]));
      asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(s1) : "v"(win[y][1]), "s"(kv[y]));
      asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(s2) : "v"(win[y][2]), "s"(kv[y]));
      asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(s3) : "v"(win[y][3]), "s"(kv[y]));
    }
    // (written once by this kernel: nontemporal -- 24 -> 1024 B=256 109.9 -> 106.2 us, 24 -> 336 40.6 -> 36.6, and the
    // marginals kernel that reads the mask next is not slower for it)
    if (q < nq) __builtin_nontemporal_store(pil_clip8x4(s0, s1, s2, s3), reinterpret_cast<u32_una*>(orow));
    else if (UA && tail_bytes) {
      const uint32_t v = pil_clip8x4(s0, s1, s2, s3);
      for (int j = 0; j < tail_bytes; ++j) orow[j] = (uint8_t)(v >> (8 * j));
    }
    orow += out_w;
  }
}


}  // namespace attwarp
