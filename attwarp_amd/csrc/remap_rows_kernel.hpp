// K7 fast path: float32 bilinear resample with separable monotone maps at (close to) the HBM
// roofline.  Replaces cv2.remap in AGW/new_method.py:268-271 / MN/checkpoint_utils.py:195-198.
// The kernel template lives in this header so that the two arithmetic modes are compiled in separate
// translation units (remap_rows.hip: EXACT, remap_rows_cv2.hip: CV2).
//
// Why this shape.  The maps are separable (map_x depends on x only, map_y on y only) and
// non-decreasing, so (1) an output row needs exactly two source rows, and consecutive output rows
// need the same or the next source rows; (2) every output row of an image uses the same column taps.
// One workgroup (256 threads = 4 waves) owns a block of R consecutive output rows of one image:
//
//   HBM --16-B coalesced loads--> registers (two register sets = a 2-entry row cache; each thread owns
//   the same float4 columns of every row) --> LDS row buffer, double buffered
//   --horizontal gather (taps precomputed once per block in VGPRs)--> coalesced 256-B-per-wave stores --> HBM.
//
// EXACT mode stages ONE row per output row, the vertical lerp of the two source rows (2 x ds_read_b32 per
// output).  CV2 mode cannot: OpenCV's float path evaluates ((p00*w00 + p01*w01) + p10*w10) + p11*w11 with the
// table weights w = (1-ty|ty)*(1-tx|tx) (exact in float32 because t = k/32), which two nested lerps do not
// reproduce bit for bit.  It therefore stages BOTH source rows unblended, [top | bottom] exactly one row
// buffer apart, so that the (p00, p10) pair of a tap is one ds_read2st64_b32 (two dwords a multiple of 256
// bytes apart with one address register): 2 LDS instructions per output as in EXACT mode, twice the LDS
// footprint, no vertical lerp.
//
// Every source row of the block is read from HBM once, the block re-reads at most one halo row of its
// neighbour (1/R extra), and the XCD-aware block order makes that halo an L2 hit.  Algorithmic bytes per
// image = 2*S*S*C*4; no MFMA: there is no contraction here, the kernel is HBM-bound.
//
// Arithmetic is identical to remap_gather_kernel / the oracle in both modes.
#pragma once
#include "common.hpp"
#include "axis_blocks.hpp"
#include "attn_f32v.hpp"
#include <algorithm>

// Source-row loads and output stores.  Tuning flavour only (test hook "remap_nt"): bit 0 = NONTEMPORAL loads of the source
// rows, bit 1 = nontemporal stores.  Measured on MI355X, 1024x1024x3 float32 B=256 (rounds 2-3, docs/experiments.md): never
// a gain worth a lease-dependent loss; -15 % on a batch that fits the Infinity Cache.  The product library always issues
// plain loads and stores.
// (the register sets are clang vectors, not HIP's float4 struct: struct copies are memcpy's, and the row loop's control flow
// left an array of them in scratch memory)
typedef float rows_v4f __attribute__((ext_vector_type(4)));
// the same four floats at any 4-byte boundary (UA: rows whose length is not a multiple of 4 floats -- 683 x 3)
typedef float rows_v4f_a4 __attribute__((ext_vector_type(4), aligned(4)));
#define ATTWARP_ROW_V4(ptr_) (UA ? rows_v4f(*reinterpret_cast<const rows_v4f_a4*>(ptr_)) : *reinterpret_cast<const rows_v4f*>(ptr_))
#ifdef ATTWARP_TUNING
#define ATTWARP_ROW_STORE(ptr_, v_)                                                                                 \
  do { if (p.nt_loads & 2) __builtin_nontemporal_store((v_), (ptr_)); else *(ptr_) = (v_); } while (0)
#define ATTWARP_ROW_LOAD(X, rp_)                                                                                    \
  if (p.nt_loads & 1) {                                                                                             \
    _Pragma("unroll") for (int k = 0; k < KI; ++k) X[k] = __builtin_nontemporal_load(reinterpret_cast<const rows_v4f*>(rp_ + goff[k])); \
  } else {                                                                                                          \
    _Pragma("unroll") for (int k = 0; k < KI; ++k) X[k] = ATTWARP_ROW_V4(rp_ + goff[k]);                             \
  }
#else
#define ATTWARP_ROW_STORE(ptr_, v_) (*(ptr_) = (v_))
#define ATTWARP_ROW_LOAD(X, rp_)                                                                                    \
  { _Pragma("unroll") for (int k = 0; k < KI; ++k) X[k] = ATTWARP_ROW_V4(rp_ + goff[k]); }
#endif

namespace attwarp {

struct Taps {
  int i0, i1;
  float f;
};
__device__ __forceinline__ Taps rtaps_exact(float m, int size) {
  const float mc = clamp_coord(m, size);
  const float fl = floorf(mc);
  Taps t;
  t.f = fsub(mc, fl);
  const int i = (int)fl;
  t.i0 = min(max(i, 0), size - 1);
  t.i1 = min(max(i + 1, 0), size - 1);
  return t;
}
// OpenCV's INTER_BITS=5 coordinate quantisation: q = cvRound(m*32), index q>>5, fraction (q&31)/32.
__device__ __forceinline__ Taps rtaps_cv2(float m, int size) {
  const int q = cv_round_q5(m);
  const int i = q >> 5;
  Taps t;
  t.f = (float)(q & 31) * 0.03125f;
  t.i0 = min(max(i, 0), size - 1);
  t.i1 = min(max(i + 1, 0), size - 1);
  return t;
}
template <int MODE>
__device__ __forceinline__ Taps rtaps(float m, int size) {
  return MODE == ATTWARP_CV2 ? rtaps_cv2(m, size) : rtaps_exact(m, size);
}
// OpenCV float path: ((p00*w00 + p01*w01) + p10*w10) + p11*w11, every operation rounded (same as blend<float,CV2>
// of remap.hip; the weights are products of k/32 fractions and exact).
__device__ __forceinline__ float cv2_sum(float p00, float p01, float p10, float p11, float w00, float w01, float w10,
                                         float w11) {
  return fadd(fadd(fadd(fmul(p00, w00), fmul(p01, w01)), fmul(p10, w10)), fmul(p11, w11));
}
// The same sum with the (top, bottom) pairs as the two halves of packed float32 operations (v_pk_mul_f32 rounds each
// half like v_mul_f32): weights (oy, fy) * ox and (oy, fy) * fx, products (p00, p10) * (w00, w10) and
// (p01, p11) * (w01, w11), then the three ordered adds -- 7 VALU instructions instead of 11 per output.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float cv2_sum_pk(v2f p0 /* p00, p10 */, v2f p1 /* p01, p11 */, v2f wy /* oy, fy */, float ox,
                                            float fx) {
  const v2f w0 = wy * ox, w1 = wy * fx;          // (w00, w10), (w01, w11)
  const v2f s0 = p0 * w0, s1 = p1 * w1;          // (s00, s10), (s01, s11)
  return fadd(fadd(fadd(s0.x, s1.x), s0.y), s1.y);
}

// Integer divisions of the per-workgroup prologue without the ~25-instruction run-time divide: the channel stride is
// 1..4 (constant divisors: shifts / one multiply-high), the plane of a virtual-row element is found by comparison
// (at most 4 planes).  The prologue is a third of the kernel's VALU instructions at 336x336x3.
__device__ __forceinline__ unsigned div_small(unsigned r, int d) {
  switch (d) {
    case 1: return r;
    case 2: return r >> 1;
    case 3: return r / 3u;
    case 4: return r >> 2;
    default: return r / (unsigned)d;
  }
}
__device__ __forceinline__ int plane_of(int e, int len) {     // e / len for e < 4 * len
  return (e >= len) + (e >= 2 * len) + (e >= 3 * len);
}

struct RowsParams {
  const float* src;
  float* dst;
  const float* mx;  // [B, Wo]
  const float* my;  // [B, Ho]
  int H, W, Ho, Wo;
  int NP, CS;        // planes per image, channel stride inside a row (HWC: 1,C ; CHW: C,1)
  int row_len;       // W*CS   floats per source row of one plane
  int orow_len;      // Wo*CS
  int VLV;           // NP*row_len/4  float4 per "virtual" source row (all planes)
  int OVL;           // NP*orow_len   output floats per virtual row
  long long img_stride, plane_stride, oimg_stride, oplane_stride;  // in floats
  int R;             // output rows per block
  int nblk;          // row blocks per image
  int wpi;           // workgroups per image (and column tile): workgroup j owns row blocks j, j + wpi, j + 2 wpi, ...
  int nblocks;       // total
  int no_swz;        // block order: 0 = contiguous range per XCD, 1 = plain, g >= 2 = XCDs interleaved in groups of g blocks
#ifdef ATTWARP_TUNING  // measurement knobs of the tuning flavour (attwarp_debug_set); constants in the product library
  int alt_dir;       // 1: odd row blocks sweep bottom-up so both neighbours meet at the shared halo rows
  int lds_pad;       // extra dynamic LDS bytes (occupancy experiments)
  int skew;          // block order 0: XCD x starts x * skew blocks into its contiguous range
  int nt_loads;      // bit 0: nontemporal loads of the source rows, bit 1: nontemporal stores
  int bound;         // TUNE_BOUND bit 0: stage only the top row, 3-row LDS pool (upper bound of a three-slot ring; garbage output)
  unsigned long long* trace;   // block timeline (common.hpp: trace_buffer): phase cycles of this block go to words 7..10
#else
  static constexpr int alt_dir = 1, lds_pad = 0, skew = 0, nt_loads = 0, bound = 0;
#endif
  int map_div;       // maps belong to image b / map_div (planes of a planar image dispatched as images)
  int ntiles;        // TILED: column tiles per row (each KO*NT output elements), else 1
  int unaligned;     // rows / images that are not 16-byte aligned: the UA instantiations (remap_rows_ua.hip)
};

constexpr int RMAX = 64;
constexpr int NT_BIG = 256;    // threads per workgroup (4 waves share a row)

// Per output row the kernel issues, per thread: KI x (3 lerps x 4) vertical blend + KI ds_write_b128 (CV2: 2 KI
// ds_write_b128, no blend), one barrier, KO x (2 unpack + 2 LDS reads + lerp / 4-term sum + 1 global_store_dword).
// Everything that does not depend on the row (taps, store offsets) lives in registers, packed to keep the
// allocation low enough for >= 3-4 resident workgroups per CU: bytes in flight per CU, not ALU, bound this kernel.
//
// Source rows live in TWO register sets X0/X1 used as a 2-entry row cache.  Right after a row's staging has
// consumed the registers, the rows the NEXT output row needs are worked out (block-uniform scalar code) and
// any missing one is loaded into the set that became dead -- the load then flies during this row's barrier, LDS
// gather and stores.  Which set holds the top row is encoded in the position in the code (the row loop exists
// twice, see the row loop below), not in run-time tags: a tag-selected destination compiles to a load plus
// selects that wait for the data at once.  Correct for arbitrary (also non-monotone) maps: whatever the next row
// needs and the sets do not hold is loaded (both rows, if need be).
// AFF: output offsets are tid*4 + a block-uniform term per k (OVL == KO*256 exactly; for planar images also
// Wo % 256 == 0 so that a k-slice never straddles two planes): no per-element offset table in VGPRs.
// (forcing >= 4 waves per SIMD on the planar variants, which allocate 130-138 VGPRs, was measured: the
//  register-limited code is 4-8 % slower than running them at 3 workgroups per CU)
// TILED (interleaved / one-plane rows wider than the 4096-float LDS row): a workgroup owns a COLUMN TILE of KO*NT
// output elements of its rows.  The source span the tile needs, [min tap, max tap], is found with a block
// reduction; it is staged relative to its 4-float-aligned start, so everything after the prologue is the same
// code.  A tile whose span exceeds KI*NT*4 floats (a map that minifies more than ~1.3x inside the tile) falls back
// to direct global taps for that tile only.
// SINGLE (CV2, rows wider than 12 KB, and the one-launch step's 8-12 KB rows): one [top | bottom] buffer and two
// barriers per row instead of two buffers and one barrier (half the LDS; two buffers of 16 KB rows would not fit 64 KB).
// UA ("unaligned", interleaved rows that fit the LDS row): a row length that is not a multiple of 4 floats, or an image
// that does not start on a 16-byte boundary.  Rows still start on float boundaries, so the loads stay 16-byte loads
// relative to the ROW start (unaligned access mode); the row's last vector is loaded END-aligned -- the four floats that
// end with the row: nothing behind the image is read -- and staged at its own float index in the LDS row (it overlaps the
// vector in front of it with the same values), so the gather needs no change at all.
template <int NT, int KI, int KO, bool HWC, bool AFF, bool TILED, int MODE, bool SINGLE, bool UA = false>
__device__ __forceinline__ void remap_rows_block(const RowsParams& p, const int block_index, float* smem) {
  static_assert(!UA || (HWC && !TILED && !AFF), "unaligned rows: interleaved, not column-tiled, per-element output offsets");
  constexpr bool CV = MODE == ATTWARP_CV2;
  float* s_my = smem;                                   // RMAX floats
  constexpr int ROWF = KI * NT * 4;                     // floats per staged source row (padded to whole waves)
  constexpr int BUF = CV ? 2 * ROWF : ROWF;             // floats per LDS buffer (CV2: top row, then bottom row)
  float* rows0 = smem + RMAX;                           // two buffers, addressed with immediates
  float* rows1 = SINGLE ? rows0 : rows0 + ((CV && p.bound) ? ROWF : BUF);
  const int tid = threadIdx.x;

  // XCD-aware block order: blocks bid, bid+8, ... share an XCD (and its L2); hand each XCD a
  // contiguous range of (image, row-block) pairs so neighbouring row blocks hit the same L2.
  int bid = block_index;
  {
    const int n = p.nblocks, q = n >> 3, r = n & 7, xcd = bid & 7, idx = bid >> 3;
    if (p.no_swz == 0) {
      const int len = (xcd < r) ? q + 1 : q, start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
      int j = idx + xcd * p.skew;                          // XCD x starts x * skew blocks into its range (wraps)
      if (p.skew) j %= len;
      bid = start + j;
    }
    else if (p.no_swz >= 2) {
      // XCD x owns every 8th GROUP of g consecutive row blocks (g = no_swz): neighbouring blocks of a group still share
      // an L2 (halo hits for g-1 of g seams), while the eight XCDs work in one compact window of memory
      const int g = p.no_swz, per = 8 * g;
      const int grp = idx / g, within = idx - grp * g;
      const int cand = grp * per + xcd * g + within;
      bid = cand < (n / per) * per ? cand : bid;          // the ragged tail keeps the plain order
    }
  }
  int b, rb0, tile = 0;
  if (TILED) {          // (image, tile, row block): row blocks of one tile stay neighbours (halo rows meet in L2)
    const int per_img = p.wpi * p.ntiles;
    b = bid / per_img;
    const int rem = bid - b * per_img;
    tile = rem / p.wpi;
    rb0 = rem - tile * p.wpi;
  } else {
    b = bid / p.wpi;
    rb0 = bid - b * p.wpi;
  }

  const float* src_b = p.src + (long long)b * p.img_stride;
  float* dst_b = p.dst + (long long)b * p.oimg_stride;

  const int bm = b / p.map_div;

  // ---- what lives in registers for the whole workgroup
  unsigned pk[KO];     // LDS BYTE offset of tap 0 | tap 1 << 16   (staged rows are <= 16 KB)
  float fxr[KO];
  unsigned ooff[AFF ? 1 : KO];   // BYTE offset of the element inside an output row (incl. plane)
  unsigned goff[KI];   // BYTE offset inside a source row (incl. plane) of the float4s this thread owns
  bool direct = false; // TILED: the tile's source span does not fit the LDS row -> global taps
  unsigned f0s[TILED ? KO : 1], f1s[TILED ? KO : 1];   // TILED: absolute float indices of the two taps
  __shared__ int s_yy[RMAX];                            // output row of entry i of the workgroup's row list

  // ---- column taps, once per workgroup.  Lanes past the end of the row duplicate the last element (same value to
  //      the same address): the row loop has no per-lane branches.
  //      (macro: for rows that fit the LDS row it runs AFTER the first source rows have been requested, below)
#define ATTWARP_COLUMN_TAPS()                                                                         \
  if (TILED) {                                                                                        \
    __shared__ int s_lo[NT / WAVE], s_hi[NT / WAVE];                                                  \
    const int e0 = tile * (KO * NT), e1 = min(e0 + KO * NT, p.OVL);                                   \
    int lo = 0x7fffffff, hi = 0;                                                                      \
    _Pragma("unroll") for (int k = 0; k < KO; ++k) {                                                  \
      const int e = min(e0 + tid + NT * k, e1 - 1);                                                   \
      const int x = (int)div_small((unsigned)e, p.CS);                                                \
      const int c = e - x * p.CS;                                                                     \
      const Taps tx = rtaps<MODE>(p.mx[(long long)bm * p.Wo + x], p.W);                               \
      f0s[k] = tx.i0 * p.CS + c;                                                                      \
      f1s[k] = tx.i1 * p.CS + c;                                                                      \
      fxr[k] = tx.f;                                                                                  \
      ooff[AFF ? 0 : k] = (unsigned)e * 4u;                                                           \
      lo = min(lo, (int)min(f0s[k], f1s[k]));                                                         \
      hi = max(hi, (int)max(f0s[k], f1s[k]));                                                         \
    }                                                                                                 \
    _Pragma("unroll") for (int o = WAVE / 2; o > 0; o >>= 1) {                                        \
      lo = min(lo, __shfl_xor(lo, o, WAVE));                                                          \
      hi = max(hi, __shfl_xor(hi, o, WAVE));                                                          \
    }                                                                                                 \
    if ((tid & (WAVE - 1)) == 0) { s_lo[tid / WAVE] = lo; s_hi[tid / WAVE] = hi; }                    \
    __syncthreads();                                                                                  \
    _Pragma("unroll") for (int w = 0; w < NT / WAVE; ++w) { lo = min(lo, s_lo[w]); hi = max(hi, s_hi[w]); } \
    const int abase = lo & ~3;                          /* 4-float aligned start of the staged span */ \
    const int nf4 = (hi - abase + 4) >> 2;              /* float4s covering [abase, hi] */            \
    direct = nf4 > KI * NT;                                                                           \
    _Pragma("unroll") for (int k = 0; k < KO; ++k) pk[k] = ((f0s[k] - abase) * 4u) | (((f1s[k] - abase) * 4u) << 16); \
    _Pragma("unroll") for (int k = 0; k < KI; ++k) goff[k] = (unsigned)(abase + 4 * min(tid + NT * k, nf4 - 1)) * 4u; \
  } else {                                                                                            \
    _Pragma("unroll") for (int k = 0; k < KO; ++k) {                                                  \
      const int e = min(tid + NT * k, p.OVL - 1);                                                     \
      const int pl = HWC ? 0 : plane_of(e, p.orow_len);                                               \
      const int r = e - pl * p.orow_len;                                                              \
      const int x = (int)div_small((unsigned)r, p.CS);                                                \
      const int c = r - x * p.CS;                                                                     \
      const Taps tx = rtaps<MODE>(p.mx[(long long)bm * p.Wo + x], p.W);                               \
      const unsigned i0 = pl * p.row_len + tx.i0 * p.CS + c;                                          \
      const unsigned i1 = pl * p.row_len + tx.i1 * p.CS + c;                                          \
      pk[k] = (i0 * 4u) | ((i1 * 4u) << 16);                                                          \
      fxr[k] = tx.f;                                                                                  \
      if (!AFF) ooff[AFF ? 0 : k] = ((unsigned)(pl * p.oplane_stride) + (unsigned)r) * 4u;            \
    }                                                                                                 \
  }
  if (TILED) {
    ATTWARP_COLUMN_TAPS()
  } else {
    // ---- which float4 of a source row this thread owns (clamped: padding lanes re-read the last one)
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      const int f = min(tid + NT * k, p.VLV - 1) * 4;
      const int pl = HWC ? 0 : plane_of(f, p.row_len);
      goff[k] = ((unsigned)(pl * p.plane_stride) + (unsigned)(f - pl * p.row_len)) * 4u;
      if (UA && f + 4 > p.row_len) goff[k] = (unsigned)(p.row_len - 4) * 4u;       // the row's last vector, end-aligned
    }
  }

#ifdef ATTWARP_TUNING
#define ATTWARP_R_MARK(i_) if (p.trace && tid == 0) { __builtin_amdgcn_sched_barrier(0); \
    p.trace[TRACE_WORDS * (size_t)blockIdx.x + 11 + (i_)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
  // shader-clock cycles of a row's three phases, summed over the workgroup's rows (tools/gantt.py)
  //   [load wait + staging | look-ahead issue + barrier | gather + arithmetic + store issue]
  unsigned long long ph_stage = 0, ph_sync = 0, ph_gather = 0, ph_rows = 0;
#define ATTWARP_PHASE_T(v_) __builtin_amdgcn_sched_barrier(0); const unsigned long long v_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
#define ATTWARP_PHASE_END(a_, b_, c_)                                                               \
    { __builtin_amdgcn_sched_barrier(0); const unsigned long long e_ = __builtin_amdgcn_s_memtime(); \
      ph_stage += (b_) - (a_); ph_sync += (c_) - (b_); ph_gather += e_ - (c_); }
#else
#define ATTWARP_R_MARK(i_)
#define ATTWARP_PHASE_T(v_)
#define ATTWARP_PHASE_END(a_, b_, c_)
#endif

  // (macros, not lambdas: the register sets must stay scalar-replaced, never addressed through a pointer)
#define ATTWARP_LOAD_ROW(X, srow)                                                                   \
  do {                                                                                              \
    const char* rp_ = reinterpret_cast<const char*>(src_b + (long long)(srow) * p.row_len);         \
    ATTWARP_ROW_LOAD(X, rp_)                                                                        \
  } while (0)
  // EXACT: the vertical lerp of (XA, XC) into the row buffer; CV2: XA, then XC one row further
#define ATTWARP_BLEND(rowbuf, XA, XC, fy)                                                           \
  do {                                                                                              \
    rows_v4f* rowv_ = reinterpret_cast<rows_v4f*>(rowbuf);                                          \
    _Pragma("unroll") for (int k = 0; k < KI; ++k) {                                                \
      rows_v4f v_;                                                                                  \
      if (CV) v_ = XA[k];                                                                           \
      else {                                                                                        \
        v_.x = lerp_rn(XA[k].x, XC[k].x, fy);                                                       \
        v_.y = lerp_rn(XA[k].y, XC[k].y, fy);                                                       \
        v_.z = lerp_rn(XA[k].z, XC[k].z, fy);                                                       \
        v_.w = lerp_rn(XA[k].w, XC[k].w, fy);                                                       \
      }                                                                                             \
      if (UA && k == KI - 1) {   /* interleaved: a vector's float index in the row = its byte offset / 4 */ \
        float* at_ = reinterpret_cast<float*>(rowbuf) + (goff[k] >> 2);                             \
        *reinterpret_cast<rows_v4f_a4*>(at_) = v_;                                                  \
        if (CV) *reinterpret_cast<rows_v4f_a4*>(at_ + ROWF) = XC[k];                                \
      } else {                                                                                      \
        rowv_[tid + NT * k] = v_;                                                                   \
        if (CV && !p.bound) rowv_[ROWF / 4 + tid + NT * k] = XC[k];                                 \
      }                                                                                             \
    }                                                                                               \
  } while (0)
  // byte offset of k-slice k inside an output row (block uniform: scalar registers)
  auto kbase = [&](int k) -> unsigned {
    if (HWC) return (unsigned)(NT * 4 * k);
    const int pl = (NT * k) / p.orow_len;
    return ((unsigned)(pl * p.oplane_stride) + (unsigned)(NT * k - pl * p.orow_len)) * 4u;
  };
  // the horizontal gather of output row yabs_ from the staged rows, and its stores
#define ATTWARP_GATHER(yabs_, ty, rowbuf)                                                           \
  do {                                                                                              \
    const char* rowb = reinterpret_cast<const char*>(rowbuf);                                       \
    char* orow = reinterpret_cast<char*>(dst_b + (long long)(yabs_) * p.orow_len);                  \
    const float fy_ = ty.f, oy_ = fsub(1.0f, ty.f);                                                 \
    /* gather in parts (EXACT: 2, CV2: 3 -- four values per output live there): the LDS reads of  \
       a part in flight, then their arithmetic + stores */                                          \
    constexpr int NPART = CV ? 3 : 2;                                                               \
    _Pragma("unroll") for (int half = 0; half < NPART; ++half) {                                    \
      constexpr int KH = (KO + NPART - 1) / NPART;                                                  \
      float v0[KH], v1[KH], u0[CV ? KH : 1], u1[CV ? KH : 1];                                       \
      _Pragma("unroll") for (int kk = 0; kk < KH; ++kk) {                                           \
        const int k = half * KH + kk;                                                               \
        if (k < KO) {                                                                               \
          unsigned w = pk[k];                                                                       \
          asm volatile("" : "+v"(w)); /* keep the packed form live: no hoisted unpacked offsets */  \
          v0[kk] = *reinterpret_cast<const float*>(rowb + (w & 0xffffu));                           \
          v1[kk] = *reinterpret_cast<const float*>(rowb + (w >> 16));                               \
          if (CV) {                                                                                 \
            u0[kk] = *reinterpret_cast<const float*>(rowb + (w & 0xffffu) + ROWF * 4);              \
            u1[kk] = *reinterpret_cast<const float*>(rowb + (w >> 16) + ROWF * 4);                  \
          }                                                                                         \
        }                                                                                           \
      }                                                                                             \
      _Pragma("unroll") for (int kk = 0; kk < KH; ++kk) {                                           \
        const int k = half * KH + kk;                                                               \
        if (k < KO) {                                                                               \
          const unsigned off = AFF ? (unsigned)(tid * 4) + kbase(k) : ooff[AFF ? 0 : k];            \
          float o_;                                                                                 \
          if (CV) {                                                                                 \
            const float fx_ = fxr[k], ox_ = fsub(1.0f, fx_);                                        \
            o_ = cv2_sum_pk(v2f{v0[kk], u0[kk]}, v2f{v1[kk], u1[kk]}, v2f{oy_, fy_}, ox_, fx_);     \
          } else {                                                                                  \
            o_ = lerp_rn(v0[kk], v1[kk], fxr[k]);                                                   \
          }                                                                                         \
          ATTWARP_ROW_STORE(reinterpret_cast<float*>(orow + off), o_);                              \
        }                                                                                           \
      }                                                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                            \
    }                                                                                               \
  } while (0)

  // ---- the workgroup's output rows.  It owns the row blocks rb0, rb0 + wpi, ... of its image (wpi == nblk: exactly
  // one; the workgroups of an image still sweep it as one compact window).  Their rows form ONE list the row loop walks
  // from end to end -- the jump from one row block to the next is just a step of more than one source row, requested
  // during the previous row's gather like any other -- so the column taps, the loads of the row maps and the exposed
  // wait for the first two source rows are paid once per workgroup, not once per row block.  Odd row blocks are listed
  // bottom-up: block i ends, and block i+1 starts, at their shared halo rows at about the same time, so the second read
  // of those rows is an L2 hit instead of HBM traffic.  (Column-tiled rows keep one list per row block: their taps
  // differ per tile, not per row block, and a tile may take the direct path.)
  const int nlist_blocks = TILED ? 1 : (p.nblk - rb0 + p.wpi - 1) / p.wpi;      // row blocks per list
  int rbl = rb0;
  do {
    const int rb_last = rbl + (nlist_blocks - 1) * p.wpi;
    const int ntot = (nlist_blocks - 1) * p.R + min(p.R, p.Ho - rb_last * p.R);   // only an image's last block is short
    if (TILED && rbl != rb0) __syncthreads();   // the previous list's last gather is done with the list and the row buffers
    for (int i = tid; i < ntot; i += NT) {
      const int rbi = i / p.R, j = i - rbi * p.R, rb = rbl + rbi * p.wpi;
      const int y0 = rb * p.R, nr = min(p.R, p.Ho - y0);
      const int y = y0 + ((p.alt_dir && (rb & 1)) ? nr - 1 - j : j);
      s_yy[i] = y;
      s_my[i] = p.my[(long long)bm * p.Ho + y];
    }
    __syncthreads();
    ATTWARP_R_MARK(0)      // row maps in LDS

    if (TILED && direct) {     // block uniform: this tile's source span does not fit the staged row
      for (int q = 0; q < ntot; ++q) {
        const Taps ty = rtaps<MODE>(s_my[q], p.H);
        const float* ra = src_b + (long long)ty.i0 * p.row_len;
        const float* rc = src_b + (long long)ty.i1 * p.row_len;
        char* orow = reinterpret_cast<char*>(dst_b + (long long)s_yy[q] * p.orow_len);
        const float oy = fsub(1.0f, ty.f);
#pragma unroll
        for (int k = 0; k < KO; ++k) {
          float o_;
          if (CV) {
            const float ox = fsub(1.0f, fxr[k]);
            o_ = cv2_sum(ra[f0s[TILED ? k : 0]], ra[f1s[TILED ? k : 0]], rc[f0s[TILED ? k : 0]], rc[f1s[TILED ? k : 0]],
                         fmul(oy, ox), fmul(oy, fxr[k]), fmul(ty.f, ox), fmul(ty.f, fxr[k]));
          } else {
            const float v0 = lerp_rn(ra[f0s[TILED ? k : 0]], rc[f0s[TILED ? k : 0]], ty.f);      // vertical first, as the staged path
            const float v1 = lerp_rn(ra[f1s[TILED ? k : 0]], rc[f1s[TILED ? k : 0]], ty.f);
            o_ = lerp_rn(v0, v1, fxr[k]);
          }
          *reinterpret_cast<float*>(orow + ooff[AFF ? 0 : k]) = o_;
        }
      }
      continue;
    }

    // ---- the row loop.  The source rows of an output row live in two register sets; which set holds the TOP row is
    // not a run-time tag but the place in the code: the loop body exists twice (ATTWARP_SLOT(X0, X1): top in X0;
    // ATTWARP_SLOT(X1, X0): top in X1) and control moves from one copy to the other whenever the rows advance by one and
    // the sets swap their roles.  Every look-ahead load therefore has a destination known at compile time, its data is
    // only waited for where the NEXT row is staged (a counted `s_waitcnt vmcnt` behind this row's stores), and it is in
    // flight during this row's barrier, gather and stores.  Rounds 1-3 chose the set through tags at run time; the compiler
    // made that ONE load in front of the choice and 8 KI v_cndmask_b32 selects behind it, which need the data at once --
    // `s_waitcnt vmcnt(0)` right behind every look-ahead load, no look-ahead at all (found with the block timeline of
    // tools/gantt.py: a row's wait for its source row was the whole load latency).
    //   next row needs the same rows          nothing to load, same copy
    //   ... the next row down (or up)         ONE load into the set that became dead, the sets swap roles: other copy
    //   ... anything else (maps may be arbitrary; steps of >= 2 rows when minifying; the next row block of the list)
    //                                         both rows loaded, same copy
    rows_v4f X0[KI], X1[KI];
    {
      const Taps tf = rtaps<MODE>(s_my[0], p.H);
      ATTWARP_LOAD_ROW(X0, tf.i0);
      if (tf.i1 != tf.i0) { ATTWARP_LOAD_ROW(X1, tf.i1); } else {
#pragma unroll
        for (int k = 0; k < KI; ++k) X1[k] = rows_v4f{0.f, 0.f, 0.f, 0.f};
      }
    }
    ATTWARP_R_MARK(1)      // first rows requested
    if (!TILED) {
      ATTWARP_COLUMN_TAPS()  // (their map loads fly beside the first source rows)
    }
    ATTWARP_R_MARK(2)      // column taps done
    // (the first rows are needed at once; consuming them here, in front of the loop, leaves the loop's own waits counted:
    // `s_waitcnt vmcnt(KO + ...)` behind the previous row's stores instead of a `vmcnt(0)` that would drain them too)
#pragma unroll
    for (int k = 0; k < KI; ++k) asm volatile("" : : "v"(X0[k]), "v"(X1[k]));
    int q = 0;
    bool more = true;
#define ATTWARP_SLOT(T_, B_)                                                                        \
    for (;;) {                                                                                      \
      ATTWARP_PHASE_T(pt0_)                                                                         \
      const Taps ty = rtaps<MODE>(s_my[q], p.H);                                                    \
      const int yabs = s_yy[q];                                                                     \
      float* rowbuf = (SINGLE || !(q & 1)) ? rows0 : rows1;   /* rows alternate between the two LDS buffers: one barrier per row */ \
      if (SINGLE) __syncthreads(); /* the previous row's gather is done with the buffer */          \
      if (ty.i1 == ty.i0) ATTWARP_BLEND(rowbuf, T_, T_, ty.f); else ATTWARP_BLEND(rowbuf, T_, B_, ty.f); \
      ATTWARP_PHASE_T(pt1_)                                                                         \
      /* what the next row needs, as scalar decisions; then at most ONE conditional load per set (a set with loads in \
         several arms comes out of the compiler as a load into temporaries and copies that wait for it) */ \
      bool step = false;                                                                            \
      int rowT = -1, rowB = -1;                             /* source row to load into T_ / B_ (-1: none) */ \
      if (q + 1 < ntot) {                                                                           \
        const Taps tn = rtaps<MODE>(s_my[q + 1], p.H);                                              \
        const bool same = tn.i0 == ty.i0 && tn.i1 == ty.i1;                                         \
        const bool down = !same && tn.i0 == ty.i1 && ty.i1 != ty.i0;          /* one row down: B_ becomes the top */ \
        const bool upw = !same && !down && tn.i1 == ty.i0 && ty.i1 != ty.i0 && tn.i0 != tn.i1;   /* one row up */ \
        step = down || upw;                                                                         \
        if (down) { rowT = tn.i1 != tn.i0 ? tn.i1 : -1; }                                           \
        else if (upw) { rowB = tn.i0; }                                                             \
        else if (!same) { rowT = tn.i0; rowB = tn.i1 != tn.i0 ? tn.i1 : -1; }                       \
      }                                                                                             \
      if (rowT >= 0) ATTWARP_LOAD_ROW(T_, rowT);                                                    \
      if (rowB >= 0) ATTWARP_LOAD_ROW(B_, rowB);                                                    \
      __syncthreads();                                                                              \
      ATTWARP_PHASE_T(pt2_)                                                                         \
      ATTWARP_GATHER(yabs, ty, rowbuf);                                                             \
      ATTWARP_PHASE_END(pt0_, pt1_, pt2_)                                                           \
      if (++q >= ntot) { more = false; break; }                                                     \
      if (step) break;                                                                              \
    }
    while (more) {
      ATTWARP_SLOT(X0, X1)
      if (!more) break;
      ATTWARP_SLOT(X1, X0)
    }
#undef ATTWARP_SLOT
#ifdef ATTWARP_TUNING
    ph_rows += (unsigned long long)ntot;
#endif
  } while (TILED && (rbl += p.wpi) < p.nblk);   // row lists of this workgroup
#ifdef ATTWARP_TUNING
  if (p.trace && tid == 0) {
    unsigned long long* r = p.trace + TRACE_WORDS * (size_t)blockIdx.x;
    r[6] = 2; r[7] = ph_stage; r[8] = ph_sync; r[9] = ph_gather; r[10] = ph_rows;
  }
#endif
  ATTWARP_R_MARK(3)      // last stores issued
#undef ATTWARP_R_MARK
#undef ATTWARP_PHASE_T
#undef ATTWARP_PHASE_END
#undef ATTWARP_GATHER
#undef ATTWARP_BLEND
#undef ATTWARP_LOAD_ROW
#undef ATTWARP_COLUMN_TAPS
}

template <int NT, int KI, int KO, bool HWC, bool AFF, bool TILED, int MODE, bool SINGLE = false, bool UA = false>
__global__ __launch_bounds__(NT) void remap_rows_kernel(const RowsParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef ATTWARP_TUNING
  const TraceStart t0 = trace_now();
#endif
  remap_rows_block<NT, KI, KO, HWC, AFF, TILED, MODE, SINGLE, UA>(p, blockIdx.x, smem);
#ifdef ATTWARP_TUNING
  trace_block(p.trace, t0, 2);
#endif
}

// ---- the fused step: ONE launch = the resample of batch k + the map construction of batch k+1 + the attention
// reduce of batch k+2 (three independent pieces of work on different buffers; attwarp_warp_step_fused).  The three
// kernels of a step are 41 / 14 / 22 us at B=64 336x336: launched one behind the other -- on one stream or as graph
// branches -- every boundary costs a drain, a dispatch ramp and a few microseconds of queue hand-off; as block ranges of
// one grid there is one ramp and one tail per step, and the reduce blocks (bound by their own latency chain, not by
// HBM) drain beside the first resample blocks.  Block order: the 2B map blocks first (the longest dependent chain
// starts at once), then the reduce blocks, then the resample blocks; each range is padded to a multiple of 8 blocks so
// that block % 8 keeps naming the XCD for the resample's XCD-aware order.  Measured, 336x336x3 float32, ring of
// batches larger than the Infinity Cache (tools/ab_step.py, same-box alternating builds): B=64 0.059 ms per step
// against 0.076 (three graph branches) and 0.089 (three eager launches); B=256 0.209 against 0.228 / 0.233; reduce and
// resample chunks interleaved evenly instead: 0.064 / 0.216; 4 heads in flight per reduce wave: 0.069 / 0.233.
// A second set of buffers for the same step geometry: with nslots == 2 one launch serves TWO consecutive batches of the
// stream per piece -- R(k), R(k+1) | M(k+2), M(k+3) | A(k+4), A(k+5) -- and the ramp and the tail of the launch (about 8 us
// of a 60 us step at B=64 336x336) are paid once per two batches.
struct StepSlot2 {
  const float* src; float* dst; const float* mx; const float* my;      // R
  const void* steps; float* map_x; float* map_y;                        // M
  const void* attn; const int32_t* starts; void* out;                   // A
};
struct StepExtra {
  StepsMapsArgs maps;      // nM8 * 8 >= nslots * 2 * maps.B blocks (0: no map work)
  AttnStepArgsAny attn;    // nA reduce blocks per slot (0: none); its dtype is also the dtype of maps.steps
  int nM8, nA, nA8, nR8;   // blocks / 8 of the ranges: ceil(nslots * 2B / 8), ceil(nslots * nA / 8), ceil(nR / 8) PER SLOT
  int nslots;              // 1 or 2
  int prio;                // 1: the map blocks (one lane's dependent chain each) raise their wave priority
  StepSlot2 s1;            // slot 1 (slot 0 = the pointers of RowsParams / maps / attn)
#ifdef ATTWARP_TUNING
  unsigned long long* trace;   // block timeline (common.hpp: trace_buffer), null = off
#endif
};

// Waves per SIMD the register allocation must leave room for: the appended map / reduce blocks must not cost the
// resample blocks their occupancy (bytes in flight per CU, not ALU, bound them).  336x336x3 rows (KI = 1, KO = 4: 58
// VGPRs on its own) keep 6 waves, 1024x1024x3 (KI = 3, KO = 12: 128 VGPRs on its own) its 4 (three dwords spill; at 3
// waves the fused step is slower still: 1.30 against 1.25 ms, and against 1.22 ms for three eager launches).
constexpr int step_min_waves(int KI, int KO, bool AFF) {
  return (KI == 1 && KO == 4) ? 6 : (KI <= 2 && KO <= 8) ? 4 : (KI == 3 && KO == 12 && AFF) ? 4 : 1;
}

// one block of the step; returns the kind of work it did (0 maps, 1 reduce, 2 resample, -1 padding)
template <int NT, int KI, int KO, bool HWC, bool AFF, int MODE, bool SINGLE>
__device__ __forceinline__ int warp_step_block(const RowsParams& p, const StepExtra& ex, float* smem, float* s_tmp, float* s_pm) {
  const int blk = blockIdx.x;
  if (blk < ex.nM8 * 8) {
    if (blk >= ex.nslots * 2 * ex.maps.B) return -1;
    const bool s1 = blk >= 2 * ex.maps.B;                     // block uniform
    const int jj = blk - (s1 ? 2 * ex.maps.B : 0);
    StepsMapsArgs m = ex.maps;
    if (s1) { m.steps = ex.s1.steps; m.map_x = ex.s1.map_x; m.map_y = ex.s1.map_y; }
    if (ex.prio) __builtin_amdgcn_s_setprio(3);
    double* sd = reinterpret_cast<double*>(smem);
    if (m.step_dtype == ATTWARP_F32) axis_maps_from_steps_block<8, float>(m, jj >> 1, jj & 1, sd, s_tmp, s_pm);
    else if (m.step_dtype == ATTWARP_F16) axis_maps_from_steps_block<8, __half>(m, jj >> 1, jj & 1, sd, s_tmp, s_pm);
    else axis_maps_from_steps_block<8, __hip_bfloat16>(m, jj >> 1, jj & 1, sd, s_tmp, s_pm);
    return 0;
  }
  const int j = blk - ex.nM8 * 8;
  if (j < ex.nA8 * 8) {
    if (j >= ex.nslots * ex.nA) return -1;
    const bool s1 = j >= ex.nA;
    const int jj = j - (s1 ? ex.nA : 0);
    AttnStepArgsAny at = ex.attn;
    if (s1) { at.attn = ex.s1.attn; at.starts = ex.s1.starts; at.out = ex.s1.out; }
    if (at.dtype == ATTWARP_F32) attn_reduce_v4_block<float, 3, 2>(at.as<float>(), jj, smem);
    else if (at.dtype == ATTWARP_F16) attn_reduce_v4_block<__half, 3, 2>(at.as<__half>(), jj, smem);
    else attn_reduce_v4_block<__hip_bfloat16, 3, 2>(at.as<__hip_bfloat16>(), jj, smem);
    return 1;
  }
  // every slot's resample range is padded to a multiple of 8 blocks (and a multiple of 8 blocks precedes): block % 8 still
  // names the XCD
  int rb = j - ex.nA8 * 8;
  const bool s1 = rb >= ex.nR8 * 8;
  rb -= s1 ? ex.nR8 * 8 : 0;
  if (rb >= p.nblocks) return -1;
  RowsParams q = p;
  if (s1) { q.src = ex.s1.src; q.dst = ex.s1.dst; q.mx = ex.s1.mx; q.my = ex.s1.my; }
  remap_rows_block<NT, KI, KO, HWC, AFF, false, MODE, SINGLE>(q, rb, smem);
  return 2;
}

template <int NT, int KI, int KO, bool HWC, bool AFF, int MODE, bool SINGLE>
__global__ __launch_bounds__(NT, step_min_waves(KI, KO, AFF)) void warp_step_kernel(const RowsParams p, const StepExtra ex) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float s_tmp[64], s_pm[64];
#ifdef ATTWARP_TUNING
  const TraceStart t0 = trace_now();
  const int kind = warp_step_block<NT, KI, KO, HWC, AFF, MODE, SINGLE>(p, ex, smem, s_tmp, s_pm);
  trace_block(ex.trace, t0, kind);
#else
  warp_step_block<NT, KI, KO, HWC, AFF, MODE, SINGLE>(p, ex, smem, s_tmp, s_pm);
#endif
}

template <int MODE, bool SINGLE>
constexpr size_t rows_lds_bytes(int KI, int NT) {
  return (size_t)(RMAX + (SINGLE ? 1 : 2) * (MODE == ATTWARP_CV2 ? 2 : 1) * KI * NT * 4) * sizeof(float);
}

// FUSED selects which kernel family a translation unit instantiates: the plain resample (remap_rows.hip,
// remap_rows_cv2.hip) or the fused step (remap_step_exact.hip, remap_step_cv2.hip) -- four heavy translation units that
// compile side by side instead of two twice as long.
template <int NT, int KI, int KO, int MODE, bool SINGLE, bool FUSED>
static int launch_rows_t(const RowsParams& p, hipStream_t st, const StepExtra* ex) {
  size_t lds = rows_lds_bytes<MODE, SINGLE>(KI, NT) + (size_t)p.lds_pad;
  if (MODE == ATTWARP_CV2 && !SINGLE && p.bound) lds = (size_t)(RMAX + 3 * KI * NT * 4) * sizeof(float);
  const dim3 t(NT);
  if constexpr (FUSED) {   // warp_step_kernel: map blocks, then the reduce blocks, then the resample blocks
    if (ex->nA > 0) lds = std::max(lds, attn_v4_lds_bytes<3>());
    if (ex->nM8 > 0) lds = std::max(lds, steps_maps_lds_bytes(std::max(ex->maps.W, ex->maps.H), ex->maps.g));
    // (the one-launch step is for small images; an axis too long for the default 64 KB of LDS takes the separate launches)
    if (lds > LDS_DEFAULT_MAX)
      return fail(ATTWARP_E_UNSUPPORTED, "warp_step_fused: %zu bytes of LDS per workgroup (> %zu): use the separate launches", lds,
                  LDS_DEFAULT_MAX);
    const dim3 g((unsigned)((ex->nM8 + ex->nA8 + ex->nslots * ex->nR8) * 8));
    if (p.NP == 1 && p.OVL == KO * NT)
      hipLaunchKernelGGL((warp_step_kernel<NT, KI, KO, true, true, MODE, SINGLE>), g, t, lds, st, p, *ex);
    else if (p.NP == 1)
      hipLaunchKernelGGL((warp_step_kernel<NT, KI, KO, true, false, MODE, SINGLE>), g, t, lds, st, p, *ex);
    else if (p.OVL == KO * NT && p.orow_len % NT == 0)
      hipLaunchKernelGGL((warp_step_kernel<NT, KI, KO, false, true, MODE, SINGLE>), g, t, lds, st, p, *ex);
    else
      hipLaunchKernelGGL((warp_step_kernel<NT, KI, KO, false, false, MODE, SINGLE>), g, t, lds, st, p, *ex);
    return check_launch("warp_step_kernel");
  } else {
    const dim3 g(p.nblocks);
    if (p.NP == 1 && p.OVL == KO * NT)   // every (lane, k) is a distinct in-row element: affine store offsets
      hipLaunchKernelGGL((remap_rows_kernel<NT, KI, KO, true, true, false, MODE, SINGLE>), g, t, lds, st, p);
    else if (p.NP == 1)
      hipLaunchKernelGGL((remap_rows_kernel<NT, KI, KO, true, false, false, MODE, SINGLE>), g, t, lds, st, p);
    else if (p.OVL == KO * NT && p.orow_len % NT == 0)   // planar, every k-slice inside one plane
      hipLaunchKernelGGL((remap_rows_kernel<NT, KI, KO, false, true, false, MODE, SINGLE>), g, t, lds, st, p);
    else
      hipLaunchKernelGGL((remap_rows_kernel<NT, KI, KO, false, false, false, MODE, SINGLE>), g, t, lds, st, p);
    return check_launch("remap_rows_kernel");
  }
}

template <int NT, int KI, int MODE, bool SINGLE, bool FUSED>
static int launch_rows_ki(const RowsParams& p, int ko, hipStream_t st, const StepExtra* ex) {
  if (ko <= 4) return launch_rows_t<NT, KI, 4, MODE, SINGLE, FUSED>(p, st, ex);
  if (ko <= 8) return launch_rows_t<NT, KI, 8, MODE, SINGLE, FUSED>(p, st, ex);
  if (ko <= 12) return launch_rows_t<NT, KI, 12, MODE, SINGLE, FUSED>(p, st, ex);
  return launch_rows_t<NT, KI, 16, MODE, SINGLE, FUSED>(p, st, ex);
}

// all staged variants of one arithmetic mode; tile_ko != 0 selects the column-tiled kernel.
// KIMIN..KIMAX bounds the float4-per-thread counts this instantiation serves (the CV2 kernel with two
// [top | bottom] buffers needs 64 KB + of LDS at KI = 4, above the 64 KB a launch gets by default: KI = 4 runs
// the SINGLE-buffer form there).  FUSED: the fused step (not built for the column-tiled kernel).
template <int MODE, bool SINGLE, int KIMIN, int KIMAX, bool FUSED>
static int launch_rows_mode(const RowsParams& p, int tile_ko, hipStream_t st, const StepExtra* ex) {
  constexpr int NT = NT_BIG;
  if constexpr (FUSED) {
    if (tile_ko != 0) return fail(ATTWARP_E_UNSUPPORTED, "warp_step_fused: rows wider than 4096 floats are not fused");
  } else {
#ifdef ATTWARP_TUNING
    if (tile_ko == 4) {     // round-5 experiment: 1024-float column tiles (4 KB, like the rows of a planar batch)
      if constexpr (KIMIN <= 2 && 2 <= KIMAX) {
        const size_t lds = rows_lds_bytes<MODE, SINGLE>(2, NT) + (size_t)p.lds_pad;
        hipLaunchKernelGGL((remap_rows_kernel<NT, 2, 4, true, false, true, MODE, SINGLE>), dim3(p.nblocks), dim3(NT), lds, st, p);
        return check_launch("remap_rows_kernel");
      }
      return fail(ATTWARP_E_UNSUPPORTED, "remap_rows: tile variant not built");
    }
#endif
    if (tile_ko == 8) {
      if constexpr (KIMIN <= 3 && 3 <= KIMAX) {
        const size_t lds = rows_lds_bytes<MODE, SINGLE>(3, NT) + (size_t)p.lds_pad;
        hipLaunchKernelGGL((remap_rows_kernel<NT, 3, 8, true, false, true, MODE, SINGLE>), dim3(p.nblocks), dim3(NT), lds, st, p);
        return check_launch("remap_rows_kernel");
      }
      return fail(ATTWARP_E_UNSUPPORTED, "remap_rows: tile variant not built");
    }
    if (tile_ko == 12) {
      if constexpr (KIMIN <= 4 && 4 <= KIMAX) {
        const size_t lds = rows_lds_bytes<MODE, SINGLE>(4, NT) + (size_t)p.lds_pad;
        hipLaunchKernelGGL((remap_rows_kernel<NT, 4, 12, true, false, true, MODE, SINGLE>), dim3(p.nblocks), dim3(NT), lds, st, p);
        return check_launch("remap_rows_kernel");
      }
      return fail(ATTWARP_E_UNSUPPORTED, "remap_rows: tile variant not built");
    }
  }
  const int ki = (p.VLV + NT - 1) / NT, ko = (p.OVL + NT - 1) / NT;
  if constexpr (KIMIN <= 1 && 1 <= KIMAX) if (ki <= 1) return launch_rows_ki<NT, 1, MODE, SINGLE, FUSED>(p, ko, st, ex);
  if constexpr (KIMIN <= 2 && 2 <= KIMAX) if (ki == 2) return launch_rows_ki<NT, 2, MODE, SINGLE, FUSED>(p, ko, st, ex);
  if constexpr (KIMIN <= 3 && 3 <= KIMAX) if (ki == 3) return launch_rows_ki<NT, 3, MODE, SINGLE, FUSED>(p, ko, st, ex);
  if constexpr (KIMIN <= 4 && 4 <= KIMAX) if (ki >= 4) return launch_rows_ki<NT, 4, MODE, SINGLE, FUSED>(p, ko, st, ex);
  return fail(ATTWARP_E_UNSUPPORTED, "remap_rows: variant for %d float4 per thread not built", ki);
}

// the unaligned form (UA) of the plain resample, both modes: remap_rows_ua.hip
int launch_rows_ua(const RowsParams& p, int mode, hipStream_t st);

// defined in remap_rows.hip / remap_rows_cv2.hip (plain resample; ex == nullptr) and remap_step_exact.hip /
// remap_step_cv2.hip (the fused step; ex != nullptr)
int launch_rows_exact(const RowsParams& p, int tile_ko, hipStream_t st, const StepExtra* ex);
int launch_rows_cv2(const RowsParams& p, int tile_ko, hipStream_t st, const StepExtra* ex);
int launch_step_exact(const RowsParams& p, int tile_ko, hipStream_t st, const StepExtra* ex);
int launch_step_cv2(const RowsParams& p, int tile_ko, hipStream_t st, const StepExtra* ex);

}  // namespace attwarp
