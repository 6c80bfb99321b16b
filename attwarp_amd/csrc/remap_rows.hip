// K7 fast path: float32 bilinear resample with separable monotone maps at (close to) the HBM
// roofline.  Replaces cv2.remap in AGW/new_method.py:268-271 / MN/checkpoint_utils.py:195-198.
//
// Why this shape.  The maps are separable (map_x depends on x only, map_y on y only) and
// non-decreasing, so (1) an output row needs exactly two source rows, and consecutive output rows
// need the same or the next source rows; (2) every output row of an image uses the same column taps.
// One workgroup (256 threads = 4 waves) owns a block of R consecutive output rows of one image:
//
//   HBM --16-B coalesced loads--> registers (rows A, B and one prefetched row P; each thread owns
//   the same float4 columns of every row) --vertical lerp--> LDS (one blended row, double buffered)
//   --horizontal gather (2 x ds_read_b32 per output, taps precomputed once per block in VGPRs)-->
//   coalesced 256-B-per-wave stores --> HBM.
//
// Every source row of the block is read from HBM once (A <- B <- P slide in registers), the block
// re-reads at most one halo row of its neighbour (1/R extra), and the XCD-aware block order makes
// that halo an L2 hit.  Algorithmic bytes per image = 2*S*S*C*4; no MFMA: there is no
// contraction here, the kernel is HBM-bound.
//
// Arithmetic is identical to remap_gather_kernel / the oracle: vertical lerp first, then
// horizontal, three individually rounded float32 operations per lerp.
#include "common.hpp"

namespace attwarp {

struct Taps {
  int i0, i1;
  float f;
};
__device__ __forceinline__ Taps rtaps_exact(float m, int size) {
  const float fl = floorf(m);
  Taps t;
  t.f = fsub(m, fl);
  const float cl = fminf(fmaxf(fl, -1.0f), (float)size);
  const int i = (int)cl;
  t.i0 = min(max(i, 0), size - 1);
  t.i1 = min(max(i + 1, 0), size - 1);
  return t;
}
__device__ __forceinline__ Taps rtaps_cv2(float m, int size) {
  const float s = fminf(fmaxf(fmul(m, 32.0f), -2.0e9f), 2.0e9f);
  const int q = __float2int_rn(s);
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
  int nblk;          // blocks per image
  int nblocks;       // total
  int alt_dir;       // 1: odd row blocks sweep bottom-up so both neighbours meet at the shared halo rows
  int no_swz;        // 1: disable the XCD-aware block order (experiments)
  int lds_pad;       // extra dynamic LDS bytes (occupancy experiments)
  int map_div;       // maps belong to image b / map_div (planes of a planar image dispatched as images)
  int ntiles;        // TILED: column tiles per row (each KO*NT output elements), else 1
};

constexpr int RMAX = 64;
constexpr int NT_BIG = 256;    // threads per workgroup (4 waves share a row)

// cv2 float weights for the separable form: with t = k/32 the products (1-ty)(1-tx) ... are exact in
// float32, so  ((p00*w00 + p01*w01) + p10*w10) + p11*w11  cannot be produced by two nested lerps
// bit-for-bit.  CV2 mode therefore runs on the gather kernel; this kernel is EXACT mode only.

// Per output row the kernel issues, per thread: KI x (3 lerps x 4) vertical blend + KI ds_write_b128,
// one barrier, KO x (2 unpack + 2 ds_read_b32 + 1 lerp + 1 global_store_dword).  Everything that does
// not depend on the row (taps, store offsets) lives in registers, packed to keep the allocation low
// enough for >= 4 resident workgroups per CU: bytes in flight per CU, not ALU, bound this kernel.
//
// Source rows live in TWO register sets X0/X1 used as a 2-entry cache with tags: an output row blends
// (top, bottom) = whichever sets hold rows (i0, i1).  Right after a row's blend has consumed the
// registers, the rows the NEXT output row needs are looked up and any missing one is loaded into the
// set that became dead -- the load then flies during this row's barrier, LDS gather and stores.  The
// cache logic is correct for arbitrary (also non-monotone) maps; monotone maps simply never miss.
// AFF: output offsets are tid*4 + a block-uniform term per k (OVL == KO*256 exactly; for planar images also
// Wo % 256 == 0 so that a k-slice never straddles two planes): no per-element offset table in VGPRs.
// (forcing >= 4 waves per SIMD on the planar variants, which allocate 130-138 VGPRs, was measured: the
//  register-limited code is 4-8 % slower than running them at 3 workgroups per CU)
// TILED (interleaved / one-plane rows wider than the 4096-float LDS row): a workgroup owns a COLUMN TILE of KO*NT
// output elements of its rows.  The source span the tile needs, [min tap, max tap], is found with a block
// reduction; it is staged relative to its 4-float-aligned start, so everything after the prologue is the same
// code.  A tile whose span exceeds KI*NT*4 floats (a map that minifies more than ~1.3x inside the tile) falls back
// to direct global taps for that tile only.
template <int NT, int KI, int KO, bool HWC, bool AFF, bool TILED = false>
__global__ __launch_bounds__(NT) void remap_rows_kernel(const RowsParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_my = smem;                                   // RMAX floats
  constexpr int BUF = KI * NT * 4;                      // floats per LDS row buffer (padded to whole waves)
  float* rows0 = smem + RMAX;                           // two row buffers, addressed with immediates
  float* rows1 = rows0 + BUF;
  const int tid = threadIdx.x;

  // XCD-aware block order: blocks bid, bid+8, ... share an XCD (and its L2); hand each XCD a
  // contiguous range of (image, row-block) pairs so neighbouring row blocks hit the same L2.
  int bid = blockIdx.x;
  {
    const int n = p.nblocks, q = n >> 3, r = n & 7, xcd = bid & 7, idx = bid >> 3;
    if (!p.no_swz) bid = (xcd < r) ? xcd * (q + 1) + idx : r * (q + 1) + (xcd - r) * q + idx;
  }
  int b, rb, tile = 0;
  if (TILED) {          // (image, tile, row block): row blocks of one tile stay neighbours (halo rows meet in L2)
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
  const int y1 = min(y0 + p.R, p.Ho);
  const int nrows = y1 - y0;

  const float* src_b = p.src + (long long)b * p.img_stride;
  float* dst_b = p.dst + (long long)b * p.oimg_stride;

  const int bm = b / p.map_div;
  if (tid < nrows) s_my[tid] = p.my[(long long)bm * p.Ho + y0 + tid];

  // ---- column taps, once per block, kept in registers.  Lanes past the end of the row duplicate
  //      the last element (same value to the same address): the row loop has no per-lane branches.
  unsigned pk[KO];     // LDS BYTE offset of tap 0 | tap 1 << 16   (row buffers are <= 16 KB)
  float fxr[KO];
  unsigned ooff[AFF ? 1 : KO];   // BYTE offset of the element inside an output row (incl. plane)
  unsigned goff[KI];   // BYTE offset inside a source row (incl. plane) of the float4s this thread owns
  bool direct = false; // TILED: the tile's source span does not fit the LDS row -> global taps
  unsigned f0s[TILED ? KO : 1], f1s[TILED ? KO : 1];   // TILED: absolute float indices of the two taps
  if (TILED) {
    __shared__ int s_lo[NT / WAVE], s_hi[NT / WAVE];
    const int e0 = tile * (KO * NT), e1 = min(e0 + KO * NT, p.OVL);
    int lo = 0x7fffffff, hi = 0;
#pragma unroll
    for (int k = 0; k < KO; ++k) {
      const int e = min(e0 + tid + NT * k, e1 - 1);
      const int x = e / p.CS;
      const int c = e - x * p.CS;
      const Taps tx = rtaps<ATTWARP_EXACT>(p.mx[(long long)bm * p.Wo + x], p.W);
      f0s[k] = tx.i0 * p.CS + c;
      f1s[k] = tx.i1 * p.CS + c;
      fxr[k] = tx.f;
      ooff[k] = (unsigned)e * 4u;
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
    const int abase = lo & ~3;                          // 4-float aligned start of the staged span
    const int nf4 = (hi - abase + 4) >> 2;              // float4s covering [abase, hi]
    direct = nf4 > KI * NT;
#pragma unroll
    for (int k = 0; k < KO; ++k) pk[k] = ((f0s[k] - abase) * 4u) | (((f1s[k] - abase) * 4u) << 16);
#pragma unroll
    for (int k = 0; k < KI; ++k) goff[k] = (unsigned)(abase + 4 * min(tid + NT * k, nf4 - 1)) * 4u;
  } else {
#pragma unroll
  for (int k = 0; k < KO; ++k) {
    const int e = min(tid + NT * k, p.OVL - 1);
    const int pl = HWC ? 0 : e / p.orow_len;
    const int r = e - pl * p.orow_len;
    const int x = r / p.CS;
    const int c = r - x * p.CS;
    const Taps tx = rtaps<ATTWARP_EXACT>(p.mx[(long long)bm * p.Wo + x], p.W);
    const unsigned i0 = pl * p.row_len + tx.i0 * p.CS + c;
    const unsigned i1 = pl * p.row_len + tx.i1 * p.CS + c;
    pk[k] = (i0 * 4u) | ((i1 * 4u) << 16);
    fxr[k] = tx.f;
    if (!AFF) ooff[k] = ((unsigned)(pl * p.oplane_stride) + (unsigned)r) * 4u;
  }
  // ---- which float4 of a source row this thread owns (clamped: padding lanes re-read the last one)
#pragma unroll
  for (int k = 0; k < KI; ++k) {
    const int f = min(tid + NT * k, p.VLV - 1) * 4;
    const int pl = HWC ? 0 : f / p.row_len;
    goff[k] = ((unsigned)(pl * p.plane_stride) + (unsigned)(f - pl * p.row_len)) * 4u;
  }
  }
  __syncthreads();

  if (TILED && direct) {     // block uniform: this tile's source span does not fit the staged row
    for (int q = 0; q < nrows; ++q) {
      const Taps ty = rtaps<ATTWARP_EXACT>(s_my[q], p.H);
      const float* ra = src_b + (long long)ty.i0 * p.row_len;
      const float* rc = src_b + (long long)ty.i1 * p.row_len;
      char* orow = reinterpret_cast<char*>(dst_b + (long long)(y0 + q) * p.orow_len);
#pragma unroll
      for (int k = 0; k < KO; ++k) {
        const float v0 = lerp_rn(ra[f0s[k]], rc[f0s[k]], ty.f);      // vertical first, as the staged path
        const float v1 = lerp_rn(ra[f1s[k]], rc[f1s[k]], ty.f);
        *reinterpret_cast<float*>(orow + ooff[k]) = lerp_rn(v0, v1, fxr[k]);
      }
    }
    return;
  }

  float4 X0[KI], X1[KI];
  int t0 = -1, t1 = -1;      // which source row each register set holds (block uniform)
  // (macros, not lambdas: the register sets must stay scalar-replaced, never addressed through a pointer)
#define ATTWARP_LOAD_ROW(X, srow)                                                                   \
  do {                                                                                              \
    const char* rp_ = reinterpret_cast<const char*>(src_b + (long long)(srow) * p.row_len);         \
    _Pragma("unroll") for (int k = 0; k < KI; ++k) X[k] = *reinterpret_cast<const float4*>(rp_ + goff[k]); \
  } while (0)
#define ATTWARP_BLEND(rowbuf, XA, XC, fy)                                                           \
  do {                                                                                              \
    float4* rowv_ = reinterpret_cast<float4*>(rowbuf);                                              \
    _Pragma("unroll") for (int k = 0; k < KI; ++k) {                                                \
      float4 v_;                                                                                    \
      v_.x = lerp_rn(XA[k].x, XC[k].x, fy);                                                         \
      v_.y = lerp_rn(XA[k].y, XC[k].y, fy);                                                         \
      v_.z = lerp_rn(XA[k].z, XC[k].z, fy);                                                         \
      v_.w = lerp_rn(XA[k].w, XC[k].w, fy);                                                         \
      rowv_[tid + NT * k] = v_;                                                                     \
    }                                                                                               \
  } while (0)
  // make rows (i0, i1) resident; a set is only overwritten if it holds neither of them
#define ATTWARP_ENSURE(i0_, i1_)                                                                    \
  do {                                                                                              \
    if (t0 != (i0_) && t1 != (i0_)) {                                                               \
      if (t0 == (i1_)) { ATTWARP_LOAD_ROW(X1, i0_); t1 = (i0_); } else { ATTWARP_LOAD_ROW(X0, i0_); t0 = (i0_); } \
    }                                                                                               \
    if (t0 != (i1_) && t1 != (i1_)) {                                                               \
      if (t0 == (i0_)) { ATTWARP_LOAD_ROW(X1, i1_); t1 = (i1_); } else { ATTWARP_LOAD_ROW(X0, i1_); t0 = (i1_); } \
    }                                                                                               \
  } while (0)
  // byte offset of k-slice k inside an output row (block uniform: scalar registers)
  auto kbase = [&](int k) -> unsigned {
    if (HWC) return (unsigned)(NT * 4 * k);
    const int pl = (NT * k) / p.orow_len;
    return ((unsigned)(pl * p.oplane_stride) + (unsigned)(NT * k - pl * p.orow_len)) * 4u;
  };
  // one output row: blend, stage, prefetch for the next row, gather, store
#define ATTWARP_DO_ROW(q_, rowbuf)                                                                  \
  do {                                                                                              \
    const int yi_ = ybeg + (q_) * ystep;                                                            \
    const Taps ty = rtaps<ATTWARP_EXACT>(s_my[yi_], p.H);                                           \
    ATTWARP_ENSURE(ty.i0, ty.i1); /* no-op unless the look-ahead missed */                          \
    const bool top0 = (t0 == ty.i0);                                                                \
    const bool bot0 = (ty.i1 == ty.i0) ? top0 : (t0 == ty.i1);                                      \
    if (top0) {                                                                                     \
      if (bot0) ATTWARP_BLEND(rowbuf, X0, X0, ty.f); else ATTWARP_BLEND(rowbuf, X0, X1, ty.f);      \
    } else {                                                                                        \
      if (bot0) ATTWARP_BLEND(rowbuf, X1, X0, ty.f); else ATTWARP_BLEND(rowbuf, X1, X1, ty.f);      \
    }                                                                                               \
    if ((q_) + 1 < nrows) { /* look-ahead: fetch what the next row needs */                         \
      const Taps tn = rtaps<ATTWARP_EXACT>(s_my[yi_ + ystep], p.H);                                 \
      ATTWARP_ENSURE(tn.i0, tn.i1);                                                                 \
    }                                                                                               \
    __syncthreads();                                                                                \
    const char* rowb = reinterpret_cast<const char*>(rowbuf);                                       \
    char* orow = reinterpret_cast<char*>(dst_b + (long long)(y0 + yi_) * p.orow_len);               \
    /* gather in two halves: 2*KO/2 LDS reads in flight, then their lerps+stores (caps live registers) */ \
    _Pragma("unroll") for (int half = 0; half < 2; ++half) {                                        \
      constexpr int KH = (KO + 1) / 2;                                                              \
      float v0[KH], v1[KH];                                                                         \
      _Pragma("unroll") for (int kk = 0; kk < KH; ++kk) {                                           \
        const int k = half * KH + kk;                                                               \
        if (k < KO) {                                                                               \
          unsigned w = pk[k];                                                                       \
          asm volatile("" : "+v"(w)); /* keep the packed form live: no hoisted unpacked offsets */  \
          v0[kk] = *reinterpret_cast<const float*>(rowb + (w & 0xffffu));                           \
          v1[kk] = *reinterpret_cast<const float*>(rowb + (w >> 16));                               \
        }                                                                                           \
      }                                                                                             \
      _Pragma("unroll") for (int kk = 0; kk < KH; ++kk) {                                           \
        const int k = half * KH + kk;                                                               \
        if (k < KO) {                                                                               \
          const unsigned off = AFF ? (unsigned)(tid * 4) + kbase(k) : ooff[k];                      \
          const float o_ = lerp_rn(v0[kk], v1[kk], fxr[k]);                                         \
          *reinterpret_cast<float*>(orow + off) = o_;                                               \
        }                                                                                           \
      }                                                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                            \
    }                                                                                               \
  } while (0)

  // Odd row blocks sweep bottom-up: block i ends, and block i+1 starts, at their shared halo rows at
  // about the same time, so the second read of those rows is an L2 hit instead of HBM traffic.
  const bool up = p.alt_dir && (rb & 1);
  const int ybeg = up ? nrows - 1 : 0, ystep = up ? -1 : 1;
#pragma unroll
  for (int k = 0; k < KI; ++k) X0[k] = X1[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  {
    const Taps tf = rtaps<ATTWARP_EXACT>(s_my[ybeg], p.H);
    ATTWARP_ENSURE(tf.i0, tf.i1);
  }
  // rows alternate between the two LDS buffers (one barrier per row is enough: a thread can only be
  // one row ahead of the slowest reader, and then it writes the OTHER buffer)
  int q = 0;
  for (; q + 1 < nrows; q += 2) {
    ATTWARP_DO_ROW(q, rows0);
    ATTWARP_DO_ROW(q + 1, rows1);
  }
  if (q < nrows) ATTWARP_DO_ROW(q, rows0);
#undef ATTWARP_DO_ROW
#undef ATTWARP_ENSURE
#undef ATTWARP_BLEND
#undef ATTWARP_LOAD_ROW
}

template <int NT, int KI, int KO>
static int launch_rows_t(const RowsParams& p, hipStream_t st) {
  const size_t lds = (size_t)(RMAX + 2 * KI * NT * 4) * sizeof(float) + (size_t)p.lds_pad;
  const dim3 g(p.nblocks), t(NT);
  if (p.NP == 1 && p.OVL == KO * NT)   // every (lane, k) is a distinct in-row element: affine store offsets
    hipLaunchKernelGGL((remap_rows_kernel<NT, KI, KO, true, true>), g, t, lds, st, p);
  else if (p.NP == 1)
    hipLaunchKernelGGL((remap_rows_kernel<NT, KI, KO, true, false>), g, t, lds, st, p);
  else if (p.OVL == KO * NT && p.orow_len % NT == 0)   // planar, every k-slice inside one plane
    hipLaunchKernelGGL((remap_rows_kernel<NT, KI, KO, false, true>), g, t, lds, st, p);
  else
    hipLaunchKernelGGL((remap_rows_kernel<NT, KI, KO, false, false>), g, t, lds, st, p);
  return check_launch("remap_rows_kernel");
}

template <int NT, int KI>
static int launch_rows_ki(const RowsParams& p, int ko, hipStream_t st) {
  if (ko <= 4) return launch_rows_t<NT, KI, 4>(p, st);
  if (ko <= 8) return launch_rows_t<NT, KI, 8>(p, st);
  if (ko <= 12) return launch_rows_t<NT, KI, 12>(p, st);
  return launch_rows_t<NT, KI, 16>(p, st);
}

template <int NT>
static int launch_rows_nt(const RowsParams& p, hipStream_t st) {
  const int ki = (p.VLV + NT - 1) / NT, ko = (p.OVL + NT - 1) / NT;
  switch (ki) {
    case 1: return launch_rows_ki<NT, 1>(p, ko, st);
    case 2: return launch_rows_ki<NT, 2>(p, ko, st);
    case 3: return launch_rows_ki<NT, 3>(p, ko, st);
    default: return launch_rows_ki<NT, 4>(p, ko, st);
  }
}

// Returns via *handled whether the fast path took the request.
int launch_remap_rows(const float* src, float* dst, int layout, int B, int C, int H, int W, int Ho, int Wo,
                      const float* mx, const float* my, int mode, hipStream_t st, bool* handled) {
  *handled = false;
  if (mode != ATTWARP_EXACT) return ATTWARP_OK;
  const char* env = getenv("ATTWARP_REMAP_VARIANT");
  if (env && env[0] == 'g') return ATTWARP_OK;  // force the generic gather kernel (A/B measurements)
  RowsParams p;
  p.src = src; p.dst = dst; p.mx = mx; p.my = my;
  p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo;
  p.map_div = 1;
  // Planar images with rows of >= 3 KB: every plane is dispatched as a one-channel image of its own (the maps of
  // image b serve planes b*C .. b*C+C-1).  A workgroup then streams ONE contiguous row at a time instead of C
  // row segments H*W apart.  Measured on MI355X, [256,3,1024,1024] float32, two boxes: 1.10 / 1.22 ms against
  // 1.24 / 1.28 ms for the all-planes-per-workgroup form (5.9 / 5.3 vs 5.2 / 5.0 TB/s); the crossover is near
  // W = 700, below it the all-planes form wins (336: 252 of 256 lanes busy with three planes per row, 84 with one).
  bool split = layout == ATTWARP_CHW && C > 1 && (long long)W * 4 >= 3072 && (long long)B * C <= 2147483647LL;
  if (const char* se = getenv("ATTWARP_REMAP_CHW_SPLIT")) split = layout == ATTWARP_CHW && C > 1 && atoi(se) != 0;
  if (split) {
    p.map_div = C;
    B *= C;
    C = 1;
    layout = ATTWARP_HWC;
  }
  if (layout == ATTWARP_HWC) { p.NP = 1; p.CS = C; } else { p.NP = C; p.CS = 1; }
  p.row_len = W * p.CS;
  p.orow_len = Wo * p.CS;
  const long long VL = (long long)p.NP * p.row_len, OVL = (long long)p.NP * p.orow_len;
  // 16-byte vector loads need every plane row to start on a 16-byte boundary
  if (p.row_len % 4 != 0) return ATTWARP_OK;
  if ((reinterpret_cast<uintptr_t>(src) & 15u) != 0) return ATTWARP_OK;
  if (layout == ATTWARP_HWC ? ((long long)H * W * C) % 4 != 0 : ((long long)H * W) % 4 != 0) return ATTWARP_OK;
  // LDS indices are 16 bit and the tap tables live in VGPRs: up to 4096 floats per staged row.  Wider interleaved /
  // one-plane rows are processed in column tiles; wider multi-plane rows take the generic kernel.
  const bool tiled = VL > 4096 || OVL > 4096;
  if (tiled && (p.NP != 1 || VL > 2147483647LL / 8 || OVL > 2147483647LL / 8)) return ATTWARP_OK;
  if (const char* te = getenv("ATTWARP_REMAP_TILED")) { if (atoi(te) == 0 && tiled) return ATTWARP_OK; }
  p.VLV = (int)(VL / 4);
  p.OVL = (int)OVL;
  p.ntiles = 1;
  p.plane_stride = (layout == ATTWARP_HWC) ? 0 : (long long)H * W;
  p.oplane_stride = (layout == ATTWARP_HWC) ? 0 : (long long)Ho * Wo;
  p.img_stride = (long long)H * W * C;
  p.oimg_stride = (long long)Ho * Wo * C;
  if (p.plane_stride * p.NP > 2147483647LL || p.oplane_stride * p.NP > 2147483647LL) return ATTWARP_OK;
  // Rows per block.  Small blocks win: the set of rows being streamed at any moment is then a compact
  // window of memory (DRAM page locality), and with alternating sweep directions the halo row two
  // neighbouring blocks share is read by both at about the same time (L2 hit).  Measured on MI355X,
  // 1024x1024x3 float32, B=256: R=4 1.07 ms, R=8 1.08, R=16 1.10, R=32 1.11, R=2 1.15.
  // 2048 output elements per column tile against 3072 staged floats: maps may minify up to 1.5x inside a tile.
  // Measured, 2048x2048x3 float32 B=64 (near-identity / peaked maps): KO=8 R=8 5.1 / 5.2 TB/s, KO=12 R=8 5.2 / 4.3
  // (more tiles on the direct path), generic gather kernel 2.4.
  int TILE_KO = 8;
  if (const char* te = getenv("ATTWARP_REMAP_TILE_KO")) { const int v = atoi(te); if (v == 8 || v == 12) TILE_KO = v; }
  if (tiled) p.ntiles = (int)((OVL + TILE_KO * NT_BIG - 1) / (TILE_KO * NT_BIG));
  const long long row_bytes = tiled ? (long long)TILE_KO * NT_BIG * 4 : VL * 4;
  // (4 KB rows, 336x336x3: R=6 4.85 TB/s, R=12 4.62 at B=256; flat at B=64)
  int R = (int)((24 * 1024 + row_bytes / 2) / row_bytes);
  R = R < 4 ? 4 : (R > 16 ? 16 : R);
  if (tiled) R = 8;
  const char* renv = getenv("ATTWARP_REMAP_ROWS");
  if (renv) { int v = atoi(renv); if (v >= 1 && v <= RMAX) R = v; }
  if (R > Ho) R = Ho;
  p.R = R;
  p.nblk = (Ho + R - 1) / R;
  const long long nb = (long long)p.nblk * B * p.ntiles;
  if (nb > 2147483647LL) return ATTWARP_OK;
  p.nblocks = (int)nb;
  p.alt_dir = 1;
  if (const char* pe = getenv("ATTWARP_REMAP_ALT")) p.alt_dir = atoi(pe) != 0;
  p.no_swz = 0; p.lds_pad = 0;
  if (const char* pe = getenv("ATTWARP_REMAP_NOSWZ")) p.no_swz = atoi(pe) != 0;
  if (const char* pe = getenv("ATTWARP_REMAP_LDSPAD")) { int v = atoi(pe); if (v >= 0 && v <= 140000) p.lds_pad = v; }
  *handled = true;
  // (one-wave workgroups for rows <= 4 KB were measured too: 336x336x3, B=256: 0.136 ms vs 0.134 ms with
  //  4-wave workgroups -- no gain, so a single workgroup size is instantiated)
  if (tiled) {
    if (TILE_KO == 8) {
      const size_t lds = (size_t)(RMAX + 2 * 3 * NT_BIG * 4) * sizeof(float) + (size_t)p.lds_pad;
      hipLaunchKernelGGL((remap_rows_kernel<NT_BIG, 3, 8, true, false, true>), dim3(p.nblocks), dim3(NT_BIG), lds, st, p);
    } else {
      const size_t lds = (size_t)(RMAX + 2 * 4 * NT_BIG * 4) * sizeof(float) + (size_t)p.lds_pad;
      hipLaunchKernelGGL((remap_rows_kernel<NT_BIG, 4, 12, true, false, true>), dim3(p.nblocks), dim3(NT_BIG), lds, st, p);
    }
    return check_launch("remap_rows_kernel");
  }
  return launch_rows_nt<NT_BIG>(p, st);
}

}  // namespace attwarp
