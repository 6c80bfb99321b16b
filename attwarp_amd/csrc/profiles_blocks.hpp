// A13 on the uint8 mask of the main_batched chain (AGW/new_method.py:206-265) as device BLOCKS -- the float64 marginals
// in numpy's summation orders (profiles_u8_block) and profile -> CDF -> inverse map (attention_maps_finalize_block) --
// shared by the stand-alone kernels of profiles.hip and the one-launch chain step (chain_step.hip); plus the element
// transforms and numpy's pairwise-summation plan every kernel of profiles.hip uses.  LDS comes from the caller.
#pragma once
#include "common.hpp"
#include "interp.hpp"

namespace attwarp {

constexpr int PROF_NT = 256;

// ---- element transform applied while summing -------------------------------------------
// max(v, 0) that propagates NaN, on a float (3 float32 instructions instead of 6 float64-pair ones)
__device__ __forceinline__ float clamp_pos_f32(float v) {
  const float c = fmaxf(v, 0.0f);
  return (v != v) ? v : c;
}
struct XfClampPos {  // gt_marginals: A.clamp_min(0)
  __device__ __forceinline__ double operator()(double v) const { return (v != v) ? v : (v > 0.0 ? v : 0.0); }  // NaN propagates
  __device__ __forceinline__ double from_f32(float v) const { return (double)clamp_pos_f32(v); }
};
// (internal) the uint8 block with its 256-entry table of XfAttention<TR>(byte) handed in by the caller instead of built per
// workgroup: the one-launch chain steps serve sqrt / exp / log from a table computed ONCE per (transform, exp_scale,
// exp_divisor) by attwarp_attention_transform_lut with the same device functions -- no exp / log code in those kernels
constexpr int ATTWARP_T_LUT = 100;
template <int TR>
struct XfAttention {  // new_method: max(att,0) -> transform -> + BASE_ATTENTION   (TR = ATTWARP_T_*, compile time)
  double exp_scale, exp_divisor;
  __device__ __forceinline__ double operator()(double v) const {
    double a = (v != v) ? v : (v > 0.0 ? v : 0.0);       // np.maximum(x, 0) propagates NaN
    if (TR == ATTWARP_T_SQUARE) a = a * a;
    else if (TR == ATTWARP_T_SQRT) a = sqrt(a);            // a >= 0 or NaN here
    else if (TR == ATTWARP_T_EXP) a = exp(exp_scale * a) / exp_divisor;
    else if (TR == ATTWARP_T_LOG) a = log(a + 1e-5);
    return a + 1e-9;
  }
  // float32 input: the clamp on the float (exact), the rest as above
  __device__ __forceinline__ double from_f32(float v) const {
    double a = (double)clamp_pos_f32(v);
    if (TR == ATTWARP_T_SQUARE) a = a * a;
    else if (TR == ATTWARP_T_SQRT) a = sqrt(a);
    else if (TR == ATTWARP_T_EXP) a = exp(exp_scale * a) / exp_divisor;
    else if (TR == ATTWARP_T_LOG) a = log(a + 1e-5);
    return a + 1e-9;
  }
};

// ---- numpy's pairwise summation, restated -------------------------------------------------------
// np.sum over a contiguous axis (numpy/_core/src/umath/loops_utils.h.src, @TYPE@_pairwise_sum):
//   n < 8         : res = 0.; for i: res += a[i]
//   n <= 128      : r[0..7] = a[0..7]; r[k] += a[i+k] for i = 8, 16, ... < n - n%8;
//                   res = ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)); then res += a[i] for the n%8 tail
//   otherwise     : n2 = n/2; n2 -= n2 % 8; pairwise(a, n2) + pairwise(a+n2, n-n2)
// The recursion only depends on n, so it is flattened once on the host into a list of leaves
// (offset, length <= 128) and a postfix program over the leaf sums (0 = push next leaf, 1 = add).
// tests/test_oracle_golden.py::test_numpy_pairwise_restatement pins this against np.sum itself.
constexpr int PW_MAX_LEAVES = 128;   // rows up to 16384 elements
struct PairwisePlan {
  int nleaves;
  int nprog;
  int off[PW_MAX_LEAVES];
  int len[PW_MAX_LEAVES];
  unsigned char prog[2 * PW_MAX_LEAVES];
};

inline void pw_build_rec(int off, int n, PairwisePlan& P) {
  if (n <= 128) {
    P.off[P.nleaves] = off;
    P.len[P.nleaves] = n;
    P.nleaves++;
    P.prog[P.nprog++] = 0;
  } else {
    int n2 = n / 2;
    n2 -= n2 % 8;
    pw_build_rec(off, n2, P);
    pw_build_rec(off + n2, n - n2, P);
    P.prog[P.nprog++] = 1;
  }
}
inline bool pw_build(int n, PairwisePlan& P) {
  P.nleaves = 0;
  P.nprog = 0;
  if (n <= 0 || n > 128 * PW_MAX_LEAVES) return false;
  pw_build_rec(0, n, P);
  return true;
}
// deepest stack the postfix program reaches (<= 10 for rows up to 16384 elements)
inline int pw_depth(const PairwisePlan& P) {
  int sp = 0, d = 0;
  for (int i = 0; i < P.nprog; ++i) {
    if (P.prog[i] == 0) { ++sp; if (sp > d) d = sp; } else --sp;
  }
  return d;
}

// The same plan built on the device into LDS (by one thread; <= 2*128 steps): a plan passed by value and
// indexed dynamically is copied to scratch by the compiler (2.5 KB per lane for the two plans of the A13 finalize
// kernel, which then spent most of its 34 us on that copy).
struct PlanLds {
  int off[PW_MAX_LEAVES];
  int len[PW_MAX_LEAVES];
  unsigned char prog[2 * PW_MAX_LEAVES];
  int nleaves, nprog;
  int stack[3 * 24];          // explicit recursion stack: (offset, n, phase)
};
__device__ inline void pw_build_lds(int n, PlanLds* P) {   // call from ONE thread, then barrier
  int sp = 0, nl = 0, np = 0;
  int* st = P->stack;
  st[0] = 0; st[1] = n; st[2] = 0; sp = 1;
  while (sp > 0) {
    --sp;
    const int o = st[3 * sp], m = st[3 * sp + 1], ph = st[3 * sp + 2];
    if (ph == 1) { P->prog[np++] = 1; continue; }
    if (m <= 128) {
      P->off[nl] = o; P->len[nl] = m; ++nl;
      P->prog[np++] = 0;
      continue;
    }
    int n2 = m / 2;
    n2 -= n2 % 8;
    st[3 * sp] = o; st[3 * sp + 1] = m; st[3 * sp + 2] = 1; ++sp;                 // combine after both halves
    st[3 * sp] = o + n2; st[3 * sp + 1] = m - n2; st[3 * sp + 2] = 0; ++sp;       // right half (popped second)
    st[3 * sp] = o; st[3 * sp + 1] = n2; st[3 * sp + 2] = 0; ++sp;                // left half (popped first)
  }
  P->nleaves = nl;
  P->nprog = np;
}

// A plan copied to LDS by the workgroup that uses it (sized by its leaves, not by PW_MAX_LEAVES): a plan indexed
// dynamically straight from the kernel arguments is copied to scratch by the compiler.
struct PlanView {
  int* off;
  int* len;
  unsigned char* prog;
  int nleaves, nprog;
};
constexpr size_t plan_view_lds_bytes(int nleaves) { return (size_t)nleaves * 8 + (size_t)((2 * nleaves + 7) & ~7); }
// carve a view out of `mem` (plan_view_lds_bytes(P.nleaves) bytes, 4-byte aligned) and fill it; all threads, then barrier.
// SrcPlan: a PairwisePlan in the kernel arguments, or a plan of the same member names in global memory (the ragged chain)
template <typename SrcPlan>
__device__ __forceinline__ PlanView plan_to_lds(const SrcPlan& P, void* mem) {
  PlanView v;
  v.nleaves = P.nleaves; v.nprog = P.nprog;
  v.off = reinterpret_cast<int*>(mem);
  v.len = v.off + P.nleaves;
  v.prog = reinterpret_cast<unsigned char*>(v.len + P.nleaves);
  for (int t = threadIdx.x; t < P.nleaves; t += blockDim.x) { v.off[t] = P.off[t]; v.len[t] = P.len[t]; }
  for (int t = threadIdx.x; t < P.nprog; t += blockDim.x) v.prog[t] = P.prog[t];
  return v;
}

// Evaluate the postfix program for one row given its leaf sums (`leaf(j)` returns leaf j).
// `stack` is per-thread storage with stride `sstride` doubles (LDS), depth <= 10.
template <typename Plan, typename LeafFn>
__device__ __forceinline__ double pw_combine(const Plan& P, LeafFn leaf, double* stack, int sstride) {
  int sp = 0, next = 0;
  for (int i = 0; i < P.nprog; ++i) {
    if (P.prog[i] == 0) {
      stack[sp * sstride] = leaf(next++);
      ++sp;
    } else {
      const double r = stack[(sp - 1) * sstride], l = stack[(sp - 2) * sstride];
      stack[(sp - 2) * sstride] = l + r;
      --sp;
    }
  }
  return stack[0];
}

// np.sum(a[0..n)) in numpy's order by a whole 256-thread block (a in LDS).  Eight consecutive lanes own the
// eight strided accumulators of one leaf (32 leaves per pass), an xor-butterfly over those lanes is exactly
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), lane 0 of the group adds the leaf's tail, thread 0 runs the tree.
// leafbuf: LDS, PW_MAX_LEAVES doubles.  Result valid in ALL threads.  Ends with a barrier.
template <typename Plan>
__device__ __forceinline__ double pw_sum_block(const double* a, const Plan& P, double* leafbuf) {
  constexpr int NT = PROF_NT;
  const int tid = threadIdx.x, k = tid & 7;
  for (int l0 = 0; l0 < P.nleaves; l0 += NT / 8) {
    const int j = l0 + (tid >> 3);
    const bool live = j < P.nleaves;
    const double* x = a + (live ? P.off[j] : 0);
    const int len = live ? P.len[j] : 0;
    double r = 0.0;
    if (len >= 8) {
      r = x[k];
      for (int i = 8; i < len - (len % 8); i += 8) r += x[i + k];
    }
    r += __shfl_xor(r, 1, WAVE);
    r += __shfl_xor(r, 2, WAVE);
    r += __shfl_xor(r, 4, WAVE);
    if (live && k == 0) {
      if (len < 8) {
        r = 0.0;
        for (int i = 0; i < len; ++i) r += x[i];
      } else {
        for (int i = len - (len % 8); i < len; ++i) r += x[i];
      }
      leafbuf[j] = r;
    }
  }
  __syncthreads();
  if (tid == 0) {
    // the postfix program never needs more stack than leaves consumed: run it in place on leafbuf
    // (slot sp <= next - 1 at every push, so a push never overwrites an unread leaf)
    int sp = 0, next = 0;
    for (int i = 0; i < P.nprog; ++i) {
      if (P.prog[i] == 0) { leafbuf[sp] = leafbuf[next]; ++next; ++sp; }
      else { leafbuf[sp - 2] = leafbuf[sp - 2] + leafbuf[sp - 1]; --sp; }
    }
  }
  __syncthreads();
  const double tot = leafbuf[0];
  __syncthreads();
  return tot;
}

// ---- uint8 attention (the main_batched chain: the up-sampled mask, AGW/new_method.py:206-215) -------------------
// Same outputs as profiles_kernel<uint8_t, XfAttention<TR>> (col[b][c], ls[b][row][leaf]) in the same summation
// orders, restructured around what bounds that kernel on uint8 input (profiles/round2_chain_pmc.txt: half of the
// wave cycles parked on LDS, 26 % of the LDS cycles bank conflicts -- every byte became an 8-byte double in LDS that
// was read back twice):
//   * grid = (leaf, image) as before, but LDS holds the RAW BYTES of a band of 64 rows x the leaf's <= 128 columns,
//     double buffered through registers; the transform runs in registers where the value is consumed;
//   * the two reductions run CONCURRENTLY on different waves of the workgroup: waves 0-1 own the row sums, waves 2-3
//     the column sums (about the same number of float64 operations each);
//   * row sums: a thread owns (row, half h) and walks numpy's stride-8 accumulators k = 4h .. 4h+3 with ONE dword
//     read per 8 columns -- four independent float64 chains per thread, no cross-lane step until the final
//     ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)), one xor-1 shuffle; lane h = 0 adds the leaf's tail;
//   * column sums: a thread owns one column and adds the band's rows in ascending order (np.sum(axis=0)'s order);
//   * identity / square: (double)byte (+ exact float square) + 1e-9 in registers; sqrt / exp / log: a 256-entry
//     table of XfAttention<TR>(byte) built once per workgroup with the SAME device functions as the generic kernel
//     (bit-identical results, no per-element sqrt / exp / log).
// The row stride of the byte tile is 128 + 8 bytes = 34 dwords: the 64 dword reads of a row-sum wave (32 rows x 2
// halves) hit 64 different banks.  Requires every leaf >= 8 long.  W % 4 != 0 and any base address (UA, "unaligned":
// the up-sampled mask of a 683-pixel-wide image): leaf offsets are multiples of 8 in every plan, so only the row STARTS
// move -- the dword loads stay dwords relative to the row start (unaligned access mode) -- and only the plan's last leaf
// can end in a partial dword, which is loaded END-aligned (the four bytes that end with the row) and shifted down.
constexpr int U8_RB = 64;              // rows per band
constexpr int U8_STR = 128 + 8;        // tile row stride in bytes

// LDS (caller provided): tile 2 * U8_RB * U8_STR bytes (16-byte aligned), then lut 256 doubles for sqrt / exp / log
template <int TR>
constexpr size_t profiles_u8_lds_bytes() {
  return 2 * (size_t)U8_RB * U8_STR + ((TR == ATTWARP_T_IDENTITY || TR == ATTWARP_T_SQUARE) ? 0 : 256 * sizeof(double));
}
// img: ONE image's [H,W] bytes; (coff, len) = leaf `leaf` of the row plan of W (nleaves leaves); col: that image's [W]
// column sums, ls: its [H,nleaves] per-leaf row sums
// TR == ATTWARP_T_LUT: ext_lut = 256 doubles in global memory, XfAttention<TR>(0 .. 255) of the caller's transform (xf unused)
template <int TR, bool UA = false>
__device__ __forceinline__ void profiles_u8_block(const uint8_t* __restrict__ img, int H, int W, const XfAttention<TR>& xf,
                                                  int coff, int len, int nleaves, int leaf, double* __restrict__ col,
                                                  double* __restrict__ ls, uint8_t* lds, const double* __restrict__ ext_lut = nullptr) {
  constexpr bool ARITH = (TR == ATTWARP_T_IDENTITY || TR == ATTWARP_T_SQUARE);
  uint8_t (*tile)[U8_RB * U8_STR] = reinterpret_cast<uint8_t (*)[U8_RB * U8_STR]>(lds);
  double* lut = reinterpret_cast<double*>(lds + 2 * U8_RB * U8_STR);
  const int tid = threadIdx.x;
  const int m = len >> 3;                                          // steps of the stride-8 accumulators
  if (TR == ATTWARP_T_LUT) lut[tid] = ext_lut[tid];
  else if (!ARITH) lut[tid] = xf((double)tid);
  const uint8_t* base = img + coff;
  // element transform of a byte held as float (exact)
  auto tf = [&](float f) -> double {
    if (TR == ATTWARP_T_SQUARE) return (double)fmul(f, f) + 1e-9;   // <= 65025: exact in float32
    return (double)f + 1e-9;
  };
  // global -> registers: thread owns dword (tid & 31) of rows (tid >> 5) + 8 * pass
  const int gd = tid & 31, gr = tid >> 5;
  const int nd = UA ? (len + 3) >> 2 : len >> 2;                   // dwords per leaf row (!UA: len % 4 == 0)
  // UA: the leaf's last dword holds len % 4 bytes: loaded from the four bytes that END with the leaf, shifted down
  const bool tail = UA && 4 * gd + 4 > len && gd < nd;
  const int goff = tail ? len - 4 : 4 * gd;
  const unsigned tshr = tail ? 8u * (unsigned)(4 * gd + 4 - len) : 0u;
  constexpr int NPASS = U8_RB / 8;
  uint32_t raw[NPASS];
#define ATTWARP_U8P_FETCH(row0_)                                                                 \
  _Pragma("unroll") for (int ps = 0; ps < NPASS; ++ps) {                                          \
    const int r_ = (row0_) + gr + 8 * ps;                                                        \
    raw[ps] = (r_ < H && gd < nd) ? *reinterpret_cast<const u32_una*>(base + (size_t)r_ * W + goff) : 0u; \
    if (UA) raw[ps] >>= tshr;                                                                    \
  }
#define ATTWARP_U8P_STAGE(buf_)                                                                  \
  _Pragma("unroll") for (int ps = 0; ps < NPASS; ++ps)                                            \
      reinterpret_cast<uint32_t*>(tile[buf_] + (gr + 8 * ps) * U8_STR)[gd] = raw[ps];
  ATTWARP_U8P_FETCH(0)
  double cacc = 0.0;
  int buf = 0;
  for (int row0 = 0; row0 < H; row0 += U8_RB) {
    const int nb = min(U8_RB, H - row0);
    ATTWARP_U8P_STAGE(buf)
    __syncthreads();                                               // (also publishes lut the first time)
    if (row0 + U8_RB < H) ATTWARP_U8P_FETCH(row0 + U8_RB)
    const uint8_t* tb = tile[buf];
    if (tid < 2 * U8_RB) {
      // ---- rows (waves 0-1): thread = (row, half) ----
      const int r = tid >> 1, hh = tid & 1;
      const uint8_t* rp = tb + r * U8_STR + 4 * hh;
      // all <= 16 dwords of the chain are requested before the first add (in-row reads past the leaf are harmless:
      // the tile row is 136 bytes); steps i >= m are skipped
      uint32_t wv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) wv[i] = *reinterpret_cast<const uint32_t*>(rp + 8 * i);
      double a0, a1, a2, a3;
      if (ARITH) {
        a0 = tf((float)(wv[0] & 0xffu)); a1 = tf((float)((wv[0] >> 8) & 0xffu));
        a2 = tf((float)((wv[0] >> 16) & 0xffu)); a3 = tf((float)(wv[0] >> 24));
      } else {
        a0 = lut[wv[0] & 0xffu]; a1 = lut[(wv[0] >> 8) & 0xffu]; a2 = lut[(wv[0] >> 16) & 0xffu]; a3 = lut[wv[0] >> 24];
      }
#pragma unroll
      for (int i = 1; i < 16; ++i) {
        if (i < m) {                                               // block uniform
          if (ARITH) {
            a0 += tf((float)(wv[i] & 0xffu)); a1 += tf((float)((wv[i] >> 8) & 0xffu));
            a2 += tf((float)((wv[i] >> 16) & 0xffu)); a3 += tf((float)(wv[i] >> 24));
          } else {
            a0 += lut[wv[i] & 0xffu]; a1 += lut[(wv[i] >> 8) & 0xffu]; a2 += lut[(wv[i] >> 16) & 0xffu]; a3 += lut[wv[i] >> 24];
          }
        }
      }
      double u = (a0 + a1) + (a2 + a3);
      u = u + __shfl_xor(u, 1, WAVE);                              // ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7))
      if (hh == 0 && r < nb) {
        const uint8_t* tp = tb + r * U8_STR;
        for (int i = 8 * m; i < len; ++i) u += ARITH ? tf((float)tp[i]) : lut[tp[i]];
        ls[((size_t)row0 + r) * nleaves + leaf] = u;
      }
    } else if (tid - 2 * U8_RB < len) {
      // ---- columns (waves 2-3): ascending rows ----
      const uint8_t* cp = tb + (tid - 2 * U8_RB);
      for (int r0 = 0; r0 < nb; r0 += 16) {                          // 16 byte reads in flight, then the 16 ordered adds
        uint8_t v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = cp[(r0 + i) * U8_STR];  // rows >= nb of a partial band: stale, unused
        if (r0 + 16 <= nb) {
#pragma unroll
          for (int i = 0; i < 16; ++i) cacc = cacc + (ARITH ? tf((float)v[i]) : lut[v[i]]);
        } else {
          for (int i = 0; i < nb - r0; ++i) cacc = cacc + (ARITH ? tf((float)cp[(r0 + i) * U8_STR]) : lut[cp[(r0 + i) * U8_STR]]);
        }
      }
    }
    buf ^= 1;                                                      // the other buffer is free: its readers passed the barrier above
  }
#undef ATTWARP_U8P_FETCH
#undef ATTWARP_U8P_STAGE
  if (tid >= 2 * U8_RB && tid - 2 * U8_RB < len) col[coff + tid - 2 * U8_RB] = cacc;
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

struct MapsFinalizeArgs {
  const double* col;         // [B, w] column sums           (kernel arguments: the batch; inside the block: one image,
  const double* ls;          // [B, h, nleaves(w)] per-leaf row sums                     see maps_finalize_image)
  int h, w, new_w, new_h, transform;
  double exp_scale, exp_divisor;
  int apply_inverse;
  float* map_x;              // [B, new_w]
  float* map_y;              // [B, new_h]
  int depth_w;               // pw_depth of the row plan (the per-thread stack of the leaf-combine program)
#ifdef ATTWARP_TUNING
  unsigned long long* trace; // block timeline (common.hpp), null = off
  int bound;                 // TUNE_BOUND bits 2 / 3 (upper-bound experiments, garbage output)
#endif
};
#ifdef ATTWARP_TUNING
#define ATTWARP_F_MARK(i_) trace_mark(a.trace, i_);
#define ATTWARP_F_BOUND(bit_) (a.bound & (bit_))
#else
#define ATTWARP_F_MARK(i_)
#define ATTWARP_F_BOUND(bit_) false
#endif
// LDS of one workgroup: knots (max(h,w) + 2 doubles) | red | per-thread stacks | leaf sums | the two plans
inline size_t maps_finalize_lds_bytes(int h, int w, const PairwisePlan& Pw, const PairwisePlan& Ph) {
  const int n = h > w ? h : w, nl = Pw.nleaves > Ph.nleaves ? Pw.nleaves : Ph.nleaves;
  return (size_t)(n + 2) * 8 + (PROF_NT / WAVE) * 8 + (size_t)pw_depth(Pw) * PROF_NT * 8 + (size_t)nl * 8 +
         plan_view_lds_bytes(Pw.nleaves) + plan_view_lds_bytes(Ph.nleaves);
}
// a.col / a.ls / a.map_x / a.map_y: ONE image's column sums [w], per-leaf row sums [h,nleaves(w)] and map rows
template <typename SrcPlan>
__device__ __forceinline__ void attention_maps_finalize_block(const SrcPlan& Pw_arg, const SrcPlan& Ph_arg,
                                                              const MapsFinalizeArgs& a, int axis, double* lds) {
  constexpr int NT = PROF_NT;
  if (ATTWARP_F_BOUND(8)) return;
  const double* __restrict__ col = a.col; const double* __restrict__ ls = a.ls;
  const int h = a.h, w = a.w, new_w = a.new_w, new_h = a.new_h, transform = a.transform, apply_inverse = a.apply_inverse;
  const double exp_scale = a.exp_scale, exp_divisor = a.exp_divisor;
  float* __restrict__ map_x = a.map_x; float* __restrict__ map_y = a.map_y;
  const int nmax = h > w ? h : w, nlmax = Pw_arg.nleaves > Ph_arg.nleaves ? Pw_arg.nleaves : Ph_arg.nleaves;
  double* smem_d = lds;                                  // nmax + 2
  double* red = smem_d + nmax + 2;                       // NT / WAVE
  double* pstack = red + NT / WAVE;                      // depth_w * NT: per-thread stack of the leaf-combine program
  double* leafbuf = pstack + (size_t)a.depth_w * NT;     // nlmax
  // the two plans come from the host through the kernel arguments and are copied to LDS by all threads (one lane
  // building them with its explicit stack in LDS cost ~10 us of dependent LDS round trips per workgroup)
  unsigned char* pv = reinterpret_cast<unsigned char*>(leafbuf + nlmax);
  const PlanView Pw = plan_to_lds(Pw_arg, pv);           // numpy's pairwise plan of a row (w terms)
  const PlanView Ph = plan_to_lds(Ph_arg, pv + plan_view_lds_bytes(Pw_arg.nleaves));   // ... of a column profile (h terms)
  __syncthreads();
  ATTWARP_F_MARK(0)
  const int n = axis ? h : w;            // profile length
  const int other = axis ? w : h;        // number of terms summed into each profile entry
  const int n_out = axis ? new_h : new_w;
  const int nl = Pw.nleaves;
  float* map = axis ? map_y : map_x;
  double* xn = smem_d;                   // n+1 knots; xn[1..n] first holds the profile

  // row r of the y profile = numpy's pairwise tree over that row's leaf sums
  // (rows of <= 8 leaves, i.e. W <= 1024: the row's leaf sums are requested together -- one memory round trip per row
  //  instead of one per leaf, the leaves of a row share a 64-byte line -- and the combine program, whose leaf index is
  //  block uniform, picks them through a uniform switch; measured with tools/gantt.py, docs/experiments.md round 5)
  auto row_sum = [&](int r) -> double {
    const double* l = ls + (size_t)r * nl;
    if (nl <= 8) {
      double v0 = l[0], v1 = l[min(1, nl - 1)], v2 = l[min(2, nl - 1)], v3 = l[min(3, nl - 1)], v4 = l[min(4, nl - 1)],
             v5 = l[min(5, nl - 1)], v6 = l[min(6, nl - 1)], v7 = l[min(7, nl - 1)];
      return pw_combine(Pw, [&](int j) -> double {
        switch (__builtin_amdgcn_readfirstlane(j)) {
          case 0: return v0; case 1: return v1; case 2: return v2; case 3: return v3;
          case 4: return v4; case 5: return v5; case 6: return v6; default: return v7;
        }
      }, pstack + threadIdx.x, NT);
    }
    return pw_combine(Pw, [&](int j) { return l[j]; }, pstack + threadIdx.x, NT);
  };
  auto inv_bias = [&](double v, int terms) -> double {
    if (!apply_inverse) return v;
    return inverse_transform(v - 1e-9 * (double)terms, transform, exp_scale, exp_divisor) + 1e-9 * (double)terms;
  };
  // this axis' profile (exact numpy order), the other axis' total and the grand total (any order:
  // they only feed the `< 1e-9` fallback test and the fallback's np.mean)
  double acc_other = 0.0, all = 0.0;
  for (int k = threadIdx.x; k < w; k += blockDim.x) {
    const double v = col[k];
    if (axis == 0) xn[k + 1] = inv_bias(v, h); else acc_other += inv_bias(v, h);
  }
  for (int k = threadIdx.x; k < h; k += blockDim.x) {
    const double v = row_sum(k);
    all += v;
    if (axis == 1) xn[k + 1] = inv_bias(v, w); else acc_other += inv_bias(v, w);
  }
  ATTWARP_F_MARK(1)
  acc_other = block_sum(acc_other, red);
  all = block_sum(all, red);
  __syncthreads();
  ATTWARP_F_MARK(2)
  const double total_self = axis ? pw_sum_block(xn + 1, Ph, leafbuf) : pw_sum_block(xn + 1, Pw, leafbuf);   // np.sum(profile)
  double total = total_self;
  const bool fallback = (total_self < 1e-9) || (acc_other < 1e-9);
  ATTWARP_F_MARK(3)
  if (fallback) {
    for (int k = threadIdx.x; k < n; k += blockDim.x) xn[k + 1] = 1.0;
    // total_att_x = w * (np.mean(att_map_biased) * h); total_att_y = h * (mean * w); then max(., EPS)
    const double mean = all / ((double)h * (double)w);
    total = (double)n * (mean * (double)other);
    total = (total != total) ? total : (total > 1e-9 ? total : 1e-9);   // python max(total, EPS): NaN stays
    __syncthreads();
  }
  if (threadIdx.x == 0 && !ATTWARP_F_BOUND(4)) {
    // np.cumsum: sequential running sum (adds only; the division below is elementwise and parallel).  32 values are
    // requested from LDS before the first add of a batch: one LDS round trip per 32 dependent adds instead of per 4
    // (the lane spent most of its time waiting: 1024 knots 16 us -> 5 us).
    double c = 0.0;
    int k = 1;
    for (; k + 32 <= n + 1; k += 32) {
      double v[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) v[i] = xn[k + i];
#pragma unroll
      for (int i = 0; i < 32; ++i) { c = c + v[i]; v[i] = c; }
#pragma unroll
      for (int i = 0; i < 32; ++i) xn[k + i] = v[i];
    }
    for (; k <= n; ++k) { c = c + xn[k]; xn[k] = c; }
  }
  __syncthreads();
  ATTWARP_F_MARK(4)
  // (cum / total) * new ; knot 0 = 0 * new ; last knot = new
  for (int k = threadIdx.x + 1; k <= n; k += blockDim.x) xn[k] = (xn[k] / total) * (double)n_out;
  __syncthreads();
  if (threadIdx.x == 0) {
    xn[0] = 0.0;
    xn[n] = (double)n_out;
  }
  __syncthreads();
  ATTWARP_F_MARK(5)
  const bool mono = block_is_sorted(xn, n + 1);
  ATTWARP_F_MARK(6)
  np_interp_block(xn, n + 1, n_out, map, mono);
  ATTWARP_F_MARK(7)
}
#undef ATTWARP_F_MARK
#undef ATTWARP_F_BOUND
// the arguments of image b of a dense batch
__device__ __forceinline__ MapsFinalizeArgs maps_finalize_image(MapsFinalizeArgs a, int b, int nleaves_w) {
  a.col += (size_t)b * a.w;
  a.ls += (size_t)b * a.h * nleaves_w;
  a.map_x += (size_t)b * a.new_w;
  a.map_y += (size_t)b * a.new_h;
  return a;
}

}  // namespace attwarp
