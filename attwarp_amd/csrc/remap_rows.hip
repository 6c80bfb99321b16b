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
};

constexpr int RMAX = 64;
constexpr int NT = 256;

// cv2 float weights for the separable form: with t = k/32 the products (1-ty)(1-tx) ... are exact in
// float32, so  ((p00*w00 + p01*w01) + p10*w10) + p11*w11  cannot be produced by two nested lerps
// bit-for-bit.  CV2 mode therefore runs on the gather kernel; this kernel is EXACT mode only.

template <int KI, int KO>
__global__ __launch_bounds__(NT) void remap_rows_kernel(const RowsParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_my = smem;                                   // RMAX floats
  float* rows = smem + RMAX;                            // 2 * VLV*4 floats
  const int tid = threadIdx.x;

  // XCD-aware block order: blocks bid, bid+8, ... share an XCD (and its L2); hand each XCD a
  // contiguous range of (image, row-block) pairs so neighbouring row blocks hit the same L2.
  int bid = blockIdx.x;
  {
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
  const int VL = p.VLV * 4;

  if (tid < nrows) s_my[tid] = p.my[(long long)b * p.Ho + y0 + tid];

  // ---- column taps, once per block, kept in registers ----
  unsigned pk[KO];   // lds index of tap 0 | tap 1 << 16
  float fxr[KO];
  int ooff[KO];      // output offset of this element inside an output row (incl. plane)
#pragma unroll
  for (int k = 0; k < KO; ++k) {
    const int e = tid + NT * k;
    pk[k] = 0; fxr[k] = 0.f; ooff[k] = 0;
    if (e < p.OVL) {
      const int pl = e / p.orow_len;
      const int r = e - pl * p.orow_len;
      const int x = r / p.CS;
      const int c = r - x * p.CS;
      const Taps tx = rtaps<ATTWARP_EXACT>(p.mx[(long long)b * p.Wo + x], p.W);
      const unsigned i0 = pl * p.row_len + tx.i0 * p.CS + c;
      const unsigned i1 = pl * p.row_len + tx.i1 * p.CS + c;
      pk[k] = i0 | (i1 << 16);
      fxr[k] = tx.f;
      ooff[k] = (int)(pl * p.oplane_stride) + r;
    }
  }
  // ---- which float4 of a source row this thread owns ----
  int goff[KI];
#pragma unroll
  for (int k = 0; k < KI; ++k) {
    const int v = tid + NT * k;
    goff[k] = -1;
    if (v < p.VLV) {
      const int f = v * 4;
      const int pl = f / p.row_len;
      goff[k] = (int)(pl * p.plane_stride) + (f - pl * p.row_len);
    }
  }
  __syncthreads();

  // ---- is map_y non-decreasing over this block?  (always true for maps built by this library;
  //      arbitrary caller maps take the direct path below) ----
  int mono = 1;
  if (tid + 1 < nrows) {
    const Taps a = rtaps<ATTWARP_EXACT>(s_my[tid], p.H), c = rtaps<ATTWARP_EXACT>(s_my[tid + 1], p.H);
    mono = (c.i0 >= a.i0) && (c.i1 >= a.i1) && (a.i1 <= c.i0 || c.i0 == a.i0);
  }
  mono = __syncthreads_and(mono);

  if (!mono) {
    // direct 4-tap path, any maps
    for (int y = y0; y < y1; ++y) {
      const Taps ty = rtaps<ATTWARP_EXACT>(s_my[y - y0], p.H);
#pragma unroll
      for (int k = 0; k < KO; ++k) {
        const int e = tid + NT * k;
        if (e < p.OVL) {
          const int pl = e / p.orow_len;
          const unsigned i0 = (pk[k] & 0xffffu) - pl * p.row_len, i1 = (pk[k] >> 16) - pl * p.row_len;
          const float* sp = src_b + (long long)pl * p.plane_stride;
          const float* r0 = sp + (long long)ty.i0 * p.row_len;
          const float* r1 = sp + (long long)ty.i1 * p.row_len;
          const float v0 = lerp_rn(r0[i0], r1[i0], ty.f);
          const float v1 = lerp_rn(r0[i1], r1[i1], ty.f);
          dst_b[(long long)y * p.orow_len + ooff[k]] = lerp_rn(v0, v1, fxr[k]);
        }
      }
    }
    return;
  }

  // ---- streaming path ----
  float4 A[KI], Bv[KI], P[KI];
  int tagA = -1, tagB = -1, tagP = -1;
  // iterator over the distinct source rows the block needs, in increasing order
  int yq = 0, which = 0, last = -1;
  auto next_needed = [&]() -> int {
    while (yq < nrows) {
      const Taps t = rtaps<ATTWARP_EXACT>(s_my[yq], p.H);
      const int cand = which ? t.i1 : t.i0;
      yq += which;
      which ^= 1;
      if (cand > last) {
        last = cand;
        return cand;
      }
    }
    return -1;
  };
  auto load_row = [&](float4(&Rr)[KI], int s) {
    const float* rp = src_b + (long long)s * p.row_len;
#pragma unroll
    for (int k = 0; k < KI; ++k)
      if (goff[k] >= 0) Rr[k] = *reinterpret_cast<const float4*>(rp + goff[k]);
  };
#pragma unroll
  for (int k = 0; k < KI; ++k) A[k] = Bv[k] = P[k] = make_float4(0.f, 0.f, 0.f, 0.f);

  tagA = next_needed();
  load_row(A, tagA);
  tagB = next_needed();
  if (tagB >= 0) load_row(Bv, tagB);
  tagP = next_needed();
  if (tagP >= 0) load_row(P, tagP);

  int buf = 0;
  for (int yi = 0; yi < nrows; ++yi) {
    const Taps ty = rtaps<ATTWARP_EXACT>(s_my[yi], p.H);
    while (tagA != ty.i0 && tagB >= 0) {  // slide the register window A <- B <- P <- next row from HBM
#pragma unroll
      for (int k = 0; k < KI; ++k) {
        A[k] = Bv[k];
        Bv[k] = P[k];
      }
      tagA = tagB;
      tagB = tagP;
      tagP = next_needed();
      if (tagP >= 0) load_row(P, tagP);
    }
    float4* rowv = reinterpret_cast<float4*>(rows + buf * VL);
    const bool same = (ty.i1 == ty.i0);
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      if (goff[k] >= 0) {
        const float4 a = A[k];
        const float4 c = same ? a : Bv[k];
        float4 v;
        v.x = lerp_rn(a.x, c.x, ty.f);
        v.y = lerp_rn(a.y, c.y, ty.f);
        v.z = lerp_rn(a.z, c.z, ty.f);
        v.w = lerp_rn(a.w, c.w, ty.f);
        rowv[tid + NT * k] = v;
      }
    }
    __syncthreads();
    const float* rowf = rows + buf * VL;
    float* orow = dst_b + (long long)(y0 + yi) * p.orow_len;
#pragma unroll
    for (int k = 0; k < KO; ++k) {
      if (tid + NT * k < p.OVL) {
        const float v0 = rowf[pk[k] & 0xffffu];
        const float v1 = rowf[pk[k] >> 16];
        orow[ooff[k]] = lerp_rn(v0, v1, fxr[k]);
      }
    }
    buf ^= 1;
  }
}

template <int KI, int KO>
static int launch_rows_t(const RowsParams& p, hipStream_t st) {
  const size_t lds = (size_t)(RMAX + 2 * p.VLV * 4) * sizeof(float);
  hipLaunchKernelGGL((remap_rows_kernel<KI, KO>), dim3(p.nblocks), dim3(NT), lds, st, p);
  return check_launch("remap_rows_kernel");
}

template <int KI>
static int launch_rows_ki(const RowsParams& p, int ko, hipStream_t st) {
  if (ko <= 4) return launch_rows_t<KI, 4>(p, st);
  if (ko <= 8) return launch_rows_t<KI, 8>(p, st);
  if (ko <= 12) return launch_rows_t<KI, 12>(p, st);
  return launch_rows_t<KI, 16>(p, st);
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
  if (layout == ATTWARP_HWC) { p.NP = 1; p.CS = C; } else { p.NP = C; p.CS = 1; }
  p.row_len = W * p.CS;
  p.orow_len = Wo * p.CS;
  const long long VL = (long long)p.NP * p.row_len, OVL = (long long)p.NP * p.orow_len;
  // 16-byte vector loads need every plane row to start on a 16-byte boundary
  if (p.row_len % 4 != 0) return ATTWARP_OK;
  if ((reinterpret_cast<uintptr_t>(src) & 15u) != 0) return ATTWARP_OK;
  if (layout == ATTWARP_HWC ? ((long long)H * W * C) % 4 != 0 : ((long long)H * W) % 4 != 0) return ATTWARP_OK;
  if (VL > 4096 || OVL > 4096) return ATTWARP_OK;  // LDS indices are 16 bit, tables live in VGPRs
  p.VLV = (int)(VL / 4);
  p.OVL = (int)OVL;
  p.plane_stride = (layout == ATTWARP_HWC) ? 0 : (long long)H * W;
  p.oplane_stride = (layout == ATTWARP_HWC) ? 0 : (long long)Ho * Wo;
  p.img_stride = (long long)H * W * C;
  p.oimg_stride = (long long)Ho * Wo * C;
  if (p.plane_stride * p.NP > 2147483647LL || p.oplane_stride * p.NP > 2147483647LL) return ATTWARP_OK;
  // rows per block: enough blocks to fill 256 CUs x ~6 resident blocks a few times over, but keep
  // the halo re-read (1/R) small.
  long long total_rows = (long long)B * Ho;
  int R = (int)(total_rows / (256 * 6 * 3));
  R = R < 8 ? 8 : (R > 32 ? 32 : R);
  const char* renv = getenv("ATTWARP_REMAP_ROWS");
  if (renv) { int v = atoi(renv); if (v >= 1 && v <= RMAX) R = v; }
  if (R > Ho) R = Ho;
  p.R = R;
  p.nblk = (Ho + R - 1) / R;
  const long long nb = (long long)p.nblk * B;
  if (nb > 2147483647LL) return ATTWARP_OK;
  p.nblocks = (int)nb;
  const int ki = (p.VLV + NT - 1) / NT, ko = (p.OVL + NT - 1) / NT;
  *handled = true;
  switch (ki) {
    case 1: return launch_rows_ki<1>(p, ko, st);
    case 2: return launch_rows_ki<2>(p, ko, st);
    case 3: return launch_rows_ki<3>(p, ko, st);
    default: return launch_rows_ki<4>(p, ko, st);
  }
}

}  // namespace attwarp
