// K7, LDS-ring variant of the roofline resample kernel (float32, exact bilinear, separable maps).
//
// Same decomposition and arithmetic as remap_rows_kernel (remap_rows.hip): one 256-thread workgroup owns
// R consecutive output rows of one image, vertical lerp of the two source rows first, then the horizontal
// lerp; results are bit-identical.  What differs is WHERE source rows live:
//
//   HBM --global_load_dwordx4 into ONE fixed register set P (the row two slides ahead)-->
//       ds_write_b128 into a 3-slot LDS ring (slot = position in the block's needed-row sequence % 3)
//   output row: 4 x ds_read_b32 per element straight from the two ring rows it needs
//               (taps i0, i1 in the top and the bottom row), 3 lerps, one coalesced store.
//
// Why: in remap_rows_kernel the two cached rows alternate between two register sets depending on the
// data-dependent row sequence; the compiler resolves that with register-set copies and conservative
// waits around every conditional load, which serialises the look-ahead and costs ~25 VGPRs.  Here the
// data-dependent part is an LDS *address* (a scalar), the load always targets the same registers, and
// the code is straight-line: prefetch depth is two source rows (one landed in LDS, one in flight), there
// is no blend phase, no intermediate row, and exactly one barrier per source row consumed (none for
// output rows that reuse the same two source rows, i.e. in magnified regions).
#include "common.hpp"

namespace attwarp {

namespace ring {

struct Taps {
  int i0, i1;
  float f;
};
__device__ __forceinline__ Taps taps(float m, int size) {
  const float fl = floorf(m);
  Taps t;
  t.f = fsub(m, fl);
  const float cl = fminf(fmaxf(fl, -1.0f), (float)size);
  const int i = (int)cl;
  t.i0 = min(max(i, 0), size - 1);
  t.i1 = min(max(i + 1, 0), size - 1);
  return t;
}

struct Params {
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
  int alt_dir;       // odd row blocks sweep bottom-up (halo rows meet in time -> L2 hit)
  int lds_pad;
};

constexpr int RMAX = 64;
constexpr int NT = 256;
constexpr int NS = 3;   // ring slots

template <int KI, int KO, bool HWC, bool AFF>
__global__ __launch_bounds__(NT) void remap_ring_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_my = smem;                                   // RMAX floats
  constexpr int SLOT = KI * NT * 4;                     // floats per ring slot (padded to whole waves)
  float* ringf = smem + RMAX;
  const int tid = threadIdx.x;

  int bid = blockIdx.x;
  {  // XCD-aware block order (see remap_rows.hip)
    const int n = p.nblocks, q = n >> 3, r = n & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r) ? xcd * (q + 1) + idx : r * (q + 1) + (xcd - r) * q + idx;
  }
  const int b = bid / p.nblk;
  const int rb = bid - b * p.nblk;
  const int y0 = rb * p.R;
  const int y1 = min(y0 + p.R, p.Ho);
  const int nrows = y1 - y0;
  const float* src_b = p.src + (long long)b * p.img_stride;
  float* dst_b = p.dst + (long long)b * p.oimg_stride;

  if (tid < nrows) s_my[tid] = p.my[(long long)b * p.Ho + y0 + tid];

  // float4 of a source row owned by this thread (padding lanes re-read the last one)
  unsigned goff[KI];
#pragma unroll
  for (int k = 0; k < KI; ++k) {
    const int f = min(tid + NT * k, p.VLV - 1) * 4;
    const int pl = HWC ? 0 : f / p.row_len;
    goff[k] = ((unsigned)(pl * p.plane_stride) + (unsigned)(f - pl * p.row_len)) * 4u;
  }
  __syncthreads();

  const bool up = p.alt_dir && (rb & 1);
  const int ybeg = up ? nrows - 1 : 0, ystep = up ? -1 : 1;

  // monotone check (in processing order the needed rows must not go backwards)
  int mono = 1;
  if (tid + 1 < nrows) {
    const Taps a = taps(s_my[tid], p.H), c = taps(s_my[tid + 1], p.H);
    mono = (c.i0 >= a.i0) && (c.i1 >= a.i1);
  }
  mono = __syncthreads_and(mono);

  // ---- iterator over the distinct source rows the block needs, in processing order ----
  int sq = 0, which = 0, last = -1;   // down: ascending rows; up: descending rows (compare on negated index)
  auto next_needed = [&]() -> int {
    while (sq < nrows) {
      const Taps t = taps(s_my[ybeg + sq * ystep], p.H);
      // down: first i0 then i1 ; up: first i1 then i0
      const int cand = (which == 0) ? (up ? t.i1 : t.i0) : (up ? t.i0 : t.i1);
      sq += which;
      which ^= 1;
      const int key = up ? (p.H - cand) : (cand + 1);   // strictly increasing key in processing order
      if (key > last) {
        last = key;
        return cand;
      }
    }
    return -1;
  };

  // the one in-flight source row: named registers (an array here is not always scalar-replaced)
  float4 P0, P1, P2, P3;
  P0 = P1 = P2 = P3 = make_float4(0.f, 0.f, 0.f, 0.f);
#define RING_LOAD(srow)                                                                              \
  do {                                                                                               \
    const char* rp_ = reinterpret_cast<const char*>(src_b + (long long)(srow) * p.row_len);          \
    P0 = *reinterpret_cast<const float4*>(rp_ + goff[0]);                                            \
    if constexpr (KI > 1) P1 = *reinterpret_cast<const float4*>(rp_ + goff[KI > 1 ? 1 : 0]);         \
    if constexpr (KI > 2) P2 = *reinterpret_cast<const float4*>(rp_ + goff[KI > 2 ? 2 : 0]);         \
    if constexpr (KI > 3) P3 = *reinterpret_cast<const float4*>(rp_ + goff[KI > 3 ? 3 : 0]);         \
  } while (0)
#define RING_WRITE(slot)                                                                             \
  do {                                                                                               \
    float4* sv_ = reinterpret_cast<float4*>(ringf + (slot) * SLOT);                                  \
    sv_[tid] = P0;                                                                                   \
    if constexpr (KI > 1) sv_[tid + NT] = P1;                                                        \
    if constexpr (KI > 2) sv_[tid + 2 * NT] = P2;                                                    \
    if constexpr (KI > 3) sv_[tid + 3 * NT] = P3;                                                    \
  } while (0)

  // ---- column taps in registers (packed LDS byte offsets | fx), see remap_rows.hip ----
  unsigned pk[KO];
  float fxr[KO];
  unsigned ooff[AFF ? 1 : KO];

  // ring state.  Q = the block's needed rows in processing order; Q[j] lives in slot j % 3.
  //   qtop  : index in Q of the first row the current output row needs
  //   qfill : how many rows of Q have been written to LDS so far (window = Q[qtop .. qfill-1], <= 3 rows)
  //   pend  : Q[qfill], in flight in P (or -1 when Q is exhausted)
  int tag0 = -1, tag1 = -1, tag2 = -1;        // source row held by each slot (block uniform)
#define TAG_OF(slot) ((slot) == 0 ? tag0 : ((slot) == 1 ? tag1 : tag2))
#define SET_TAG(slot, v) do { if ((slot) == 0) tag0 = (v); else if ((slot) == 1) tag1 = (v); else tag2 = (v); } while (0)
  int qtop = 0, qfill = 0, pend = -1;

  if (mono) {
    pend = next_needed();                     // Q[0]: issue its load now, compute the taps while it flies
    RING_LOAD(pend);
  }
#pragma unroll
  for (int k = 0; k < KO; ++k) {
    const int e = min(tid + NT * k, p.OVL - 1);
    const int pl = HWC ? 0 : e / p.orow_len;
    const int r = e - pl * p.orow_len;
    const int x = r / p.CS;
    const int c = r - x * p.CS;
    const Taps tx = taps(p.mx[(long long)b * p.Wo + x], p.W);
    const unsigned i0 = pl * p.row_len + tx.i0 * p.CS + c;
    const unsigned i1 = pl * p.row_len + tx.i1 * p.CS + c;
    pk[k] = (i0 * 4u) | ((i1 * 4u) << 16);
    fxr[k] = tx.f;
    if (!AFF) ooff[k] = ((unsigned)(pl * p.oplane_stride) + (unsigned)r) * 4u;
  }

  if (!mono) {
    // arbitrary caller maps: direct 4-tap path from global memory
    for (int y = y0; y < y1; ++y) {
      const Taps ty = taps(s_my[y - y0], p.H);
#pragma unroll
      for (int k = 0; k < KO; ++k) {
        const int e = min(tid + NT * k, p.OVL - 1);
        const int pl = HWC ? 0 : e / p.orow_len;
        const unsigned i0 = ((pk[k] & 0xffffu) >> 2) - pl * p.row_len, i1 = (pk[k] >> 18) - pl * p.row_len;
        const float* sp = src_b + (long long)pl * p.plane_stride;
        const float* q0 = sp + (long long)ty.i0 * p.row_len;
        const float* q1 = sp + (long long)ty.i1 * p.row_len;
        const float v0 = lerp_rn(q0[i0], q1[i0], ty.f);
        const float v1 = lerp_rn(q0[i1], q1[i1], ty.f);
        const unsigned off = AFF ? (unsigned)(tid * 4 + NT * 4 * k) : ooff[k];
        *reinterpret_cast<float*>(reinterpret_cast<char*>(dst_b + (long long)y * p.orow_len) + off) = lerp_rn(v0, v1, fxr[k]);
      }
    }
    return;
  }

  // initial fill: Q[0..2] into slots 0..2, Q[3] left in flight
  for (int j = 0; j < NS && pend >= 0; ++j) {
    RING_WRITE(j);
    SET_TAG(j, pend);
    qfill = j + 1;
    pend = next_needed();
    if (pend >= 0) RING_LOAD(pend);
  }
  __syncthreads();

  for (int q = 0; q < nrows; ++q) {
    const int yi = ybeg + q * ystep;
    const Taps ty = taps(s_my[yi], p.H);
    const int first = up ? ty.i1 : ty.i0;        // the needed row that comes first in processing order
    // slide the window until its first row is `first`
    while (TAG_OF(qtop % NS) != first && qtop + 1 < qfill) {
      ++qtop;
      __syncthreads();   // (a) every wave is done reading the row that just left the window, so its slot may be
                         //     overwritten; (b) the row written at the previous slide becomes visible
      if (pend >= 0) {
        const int slot = qfill % NS;             // the slot that was just freed
        RING_WRITE(slot);
        SET_TAG(slot, pend);
        ++qfill;
        pend = next_needed();
        if (pend >= 0) RING_LOAD(pend);
      }
    }
    const int sa = qtop % NS;
    const int sb = (ty.i1 != ty.i0) ? (qtop + 1) % NS : sa;
    const int top_slot = up ? sb : sa, bot_slot = up ? sa : sb;   // top = source row i0, bottom = row i1
    const char* rowA = reinterpret_cast<const char*>(ringf + top_slot * SLOT);
    const char* rowB = reinterpret_cast<const char*>(ringf + bot_slot * SLOT);
    char* orow = reinterpret_cast<char*>(dst_b + (long long)(y0 + yi) * p.orow_len);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      constexpr int KH = (KO + 1) / 2;
      float a0[KH], a1[KH], c0[KH], c1[KH];
#pragma unroll
      for (int kk = 0; kk < KH; ++kk) {
        const int k = half * KH + kk;
        if (k < KO) {
          unsigned w = pk[k];
          asm volatile("" : "+v"(w));          // keep the packed form live: no hoisted unpacked offsets
          a0[kk] = *reinterpret_cast<const float*>(rowA + (w & 0xffffu));
          a1[kk] = *reinterpret_cast<const float*>(rowA + (w >> 16));
          c0[kk] = *reinterpret_cast<const float*>(rowB + (w & 0xffffu));
          c1[kk] = *reinterpret_cast<const float*>(rowB + (w >> 16));
        }
      }
#pragma unroll
      for (int kk = 0; kk < KH; ++kk) {
        const int k = half * KH + kk;
        if (k < KO) {
          const float v0 = lerp_rn(a0[kk], c0[kk], ty.f);
          const float v1 = lerp_rn(a1[kk], c1[kk], ty.f);
          const unsigned off = AFF ? (unsigned)(tid * 4 + NT * 4 * k) : ooff[k];
          *reinterpret_cast<float*>(orow + off) = lerp_rn(v0, v1, fxr[k]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#undef TAG_OF
#undef SET_TAG
#undef RING_LOAD
#undef RING_WRITE
}

template <int KI, int KO>
static int launch_t(const Params& p, hipStream_t st) {
  const size_t lds = (size_t)(RMAX + NS * KI * NT * 4) * sizeof(float) + (size_t)p.lds_pad;
  const dim3 g(p.nblocks), t(NT);
  if (p.NP == 1 && p.OVL == KO * NT)
    hipLaunchKernelGGL((remap_ring_kernel<KI, KO, true, true>), g, t, lds, st, p);
  else if (p.NP == 1)
    hipLaunchKernelGGL((remap_ring_kernel<KI, KO, true, false>), g, t, lds, st, p);
  else
    hipLaunchKernelGGL((remap_ring_kernel<KI, KO, false, false>), g, t, lds, st, p);
  return check_launch("remap_ring_kernel");
}
template <int KI>
static int launch_ki(const Params& p, int ko, hipStream_t st) {
  if (ko <= 4) return launch_t<KI, 4>(p, st);
  if (ko <= 8) return launch_t<KI, 8>(p, st);
  if (ko <= 12) return launch_t<KI, 12>(p, st);
  return launch_t<KI, 16>(p, st);
}

}  // namespace ring

int launch_remap_ring(const float* src, float* dst, int layout, int B, int C, int H, int W, int Ho, int Wo,
                      const float* mx, const float* my, int R, hipStream_t st) {
  ring::Params p;
  p.src = src; p.dst = dst; p.mx = mx; p.my = my;
  p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo;
  if (layout == ATTWARP_HWC) { p.NP = 1; p.CS = C; } else { p.NP = C; p.CS = 1; }
  p.row_len = W * p.CS;
  p.orow_len = Wo * p.CS;
  const long long VL = (long long)p.NP * p.row_len, OVL = (long long)p.NP * p.orow_len;
  p.VLV = (int)(VL / 4);
  p.OVL = (int)OVL;
  p.plane_stride = (layout == ATTWARP_HWC) ? 0 : (long long)H * W;
  p.oplane_stride = (layout == ATTWARP_HWC) ? 0 : (long long)Ho * Wo;
  p.img_stride = (long long)H * W * C;
  p.oimg_stride = (long long)Ho * Wo * C;
  p.R = R;
  p.nblk = (Ho + R - 1) / R;
  p.nblocks = p.nblk * B;
  p.alt_dir = 1;
  if (const char* pe = getenv("ATTWARP_REMAP_ALT")) p.alt_dir = atoi(pe) != 0;
  p.lds_pad = 0;
  if (const char* pe = getenv("ATTWARP_REMAP_LDSPAD")) { int v = atoi(pe); if (v >= 0 && v <= 140000) p.lds_pad = v; }
  const int ki = (p.VLV + ring::NT - 1) / ring::NT, ko = (p.OVL + ring::NT - 1) / ring::NT;
  switch (ki) {
    case 1: return ring::launch_ki<1>(p, ko, st);
    case 2: return ring::launch_ki<2>(p, ko, st);
    case 3: return ring::launch_ki<3>(p, ko, st);
    default: return ring::launch_ki<4>(p, ko, st);
  }
}

}  // namespace attwarp
