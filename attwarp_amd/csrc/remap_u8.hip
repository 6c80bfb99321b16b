// K7 for uint8 images: the resample of the main_batched chain (AGW/new_method.py:268-271 on uint8 BGR
// images, typically 336x336 -> 500x500).  Same decomposition as remap_rows_kernel (remap_rows.hip): one
// workgroup owns R consecutive output rows of one image, vertical lerp first into an LDS float row, then
// the horizontal gather.  Arithmetic is the uint8 "exact" path of remap_gather_kernel / the oracle:
// bytes -> float32, three individually rounded lerps, round half to even, clamp to [0,255].
//
// uint8 rows are small (W*C <= 4096 bytes), so the structure is simpler than the float kernel:
//   * each thread owns ONE 16-byte piece of a source row (global_load_dwordx4, 4-byte aligned); both
//     source rows of the NEXT output row are fetched while the current row is gathered (static double
//     buffering, no row cache: the re-read of a shared row is an L1/L2 hit, HBM sees every byte once);
//   * 16 lerps per thread -> 4 x ds_write_b128 into the float row;
//   * gather: one output byte per lane per k-slice (consecutive lanes -> consecutive LDS banks), four
//     neighbouring lanes are packed into one dword with DPP row shifts and every 4th lane stores it.
#include "common.hpp"

namespace attwarp {

namespace u8k {

constexpr int NT = 256;
constexpr int RMAX = 64;

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
  const uint8_t* src;
  uint8_t* dst;
  const float* mx;
  const float* my;
  int H, W, Ho, Wo;
  int NP, CS;
  int row_len, orow_len;   // bytes per plane row
  int VL, OVL;             // bytes per virtual row (all planes)
  long long img_stride, plane_stride, oimg_stride, oplane_stride;
  int R, nblk, nblocks;
};

struct __attribute__((packed, aligned(4))) U4 {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ void blend16(const U4& a, const U4& c, float fy, float* out) {
  const uint32_t wa[4] = {a.x, a.y, a.z, a.w}, wc[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float4 v;
    v.x = lerp_rn((float)(wa[i] & 0xffu), (float)(wc[i] & 0xffu), fy);
    v.y = lerp_rn((float)((wa[i] >> 8) & 0xffu), (float)((wc[i] >> 8) & 0xffu), fy);
    v.z = lerp_rn((float)((wa[i] >> 16) & 0xffu), (float)((wc[i] >> 16) & 0xffu), fy);
    v.w = lerp_rn((float)(wa[i] >> 24), (float)(wc[i] >> 24), fy);
    reinterpret_cast<float4*>(out)[i] = v;
  }
}

// row shift left by n inside a DPP row of 16 lanes: lane i reads lane i+n (0 if it leaves the row)
template <int N>
__device__ __forceinline__ uint32_t dpp_shl(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + N, 0xf, 0xf, true);
}

template <int KO, bool HWC>
__global__ __launch_bounds__(NT) void remap_rows_u8_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_my = smem;                      // RMAX
  float* row0 = smem + RMAX;               // 2 float rows of VLP floats
  const int VLP = (p.VL + 15) & ~15;
  float* row1 = row0 + VLP;
  const int tid = threadIdx.x;

  int bid = blockIdx.x;
  {
    const int n = p.nblocks, q = n >> 3, r = n & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r) ? xcd * (q + 1) + idx : r * (q + 1) + (xcd - r) * q + idx;
  }
  const int b = bid / p.nblk;
  const int rb = bid - b * p.nblk;
  const int y0 = rb * p.R;
  const int nrows = min(y0 + p.R, p.Ho) - y0;
  const uint8_t* src_b = p.src + (long long)b * p.img_stride;
  uint8_t* dst_b = p.dst + (long long)b * p.oimg_stride;

  if (tid < nrows) s_my[tid] = p.my[(long long)b * p.Ho + y0 + tid];

  // the 16-byte piece of a source row this thread owns: ppp pieces per plane, the last piece of a plane is
  // shifted back so that it ends with the plane row (overlapping pieces write identical values)
  const int ppp = (p.row_len + 15) >> 4;
  const bool loader = tid < ppp * p.NP;
  const int ppl = min(tid / ppp, p.NP - 1);
  const int pin = min((tid - ppl * ppp) * 16, p.row_len - 16);
  const int voff = ppl * p.row_len + pin;            // byte offset inside the virtual row (= float index in LDS)
  const int goff = (int)(ppl * p.plane_stride) + pin; // byte offset inside the image, row 0

  // column taps
  unsigned pk[KO];
  float fxr[KO];
  unsigned ooff[KO];
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
    ooff[k] = (unsigned)(pl * p.oplane_stride) + (unsigned)r;
  }
  __syncthreads();

  auto load_piece = [&](int srow) -> U4 {
    U4 v = {0u, 0u, 0u, 0u};
    if (loader) v = *reinterpret_cast<const U4*>(src_b + (long long)srow * p.row_len + goff);
    return v;
  };

  Taps tcur = taps(s_my[0], p.H);
  U4 A = load_piece(tcur.i0), C = load_piece(tcur.i1);
  for (int q = 0; q < nrows; ++q) {
    float* rowbuf = (q & 1) ? row1 : row0;
    if (loader) blend16(A, C, tcur.f, rowbuf + voff);
    if (q + 1 < nrows) {                             // fetch the next output row's two source rows now
      tcur = taps(s_my[q + 1], p.H);
      A = load_piece(tcur.i0);
      C = load_piece(tcur.i1);
    }
    __syncthreads();
    const char* rowb = reinterpret_cast<const char*>(rowbuf);
    uint8_t* orow = dst_b + (long long)(y0 + q) * p.orow_len;
#pragma unroll
    for (int k = 0; k < KO; ++k) {
      unsigned w = pk[k];
      asm volatile("" : "+v"(w));
      const float v0 = *reinterpret_cast<const float*>(rowb + (w & 0xffffu));
      const float v1 = *reinterpret_cast<const float*>(rowb + (w >> 16));
      const float o = lerp_rn(v0, v1, fxr[k]);
      const uint32_t r8 = (uint32_t)fminf(fmaxf(rintf(o), 0.0f), 255.0f);
      const uint32_t packed = r8 | (dpp_shl<1>(r8) << 8) | (dpp_shl<2>(r8) << 16) | (dpp_shl<3>(r8) << 24);
      if ((tid & 3) == 0 && tid + NT * k < p.OVL) *reinterpret_cast<uint32_t*>(orow + ooff[k]) = packed;
    }
  }
}

template <int KO>
static int launch_ko(const Params& p, hipStream_t st) {
  const int VLP = (p.VL + 15) & ~15;
  const size_t lds = (size_t)(RMAX + 2 * VLP) * sizeof(float);
  if (p.NP == 1)
    hipLaunchKernelGGL((remap_rows_u8_kernel<KO, true>), dim3(p.nblocks), dim3(NT), lds, st, p);
  else
    hipLaunchKernelGGL((remap_rows_u8_kernel<KO, false>), dim3(p.nblocks), dim3(NT), lds, st, p);
  return check_launch("remap_rows_u8_kernel");
}

}  // namespace u8k

// Returns via *handled whether the uint8 fast path took the request.
int launch_remap_rows_u8(const uint8_t* src, uint8_t* dst, int layout, int B, int C, int H, int W, int Ho, int Wo,
                         const float* mx, const float* my, int mode, hipStream_t st, bool* handled) {
  *handled = false;
  if (mode != ATTWARP_EXACT) return ATTWARP_OK;
  const char* env = getenv("ATTWARP_REMAP_VARIANT");
  if (env && env[0] == 'g') return ATTWARP_OK;
  u8k::Params p;
  p.src = src; p.dst = dst; p.mx = mx; p.my = my;
  p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo;
  if (layout == ATTWARP_HWC) { p.NP = 1; p.CS = C; } else { p.NP = C; p.CS = 1; }
  p.row_len = W * p.CS;
  p.orow_len = Wo * p.CS;
  const long long VL = (long long)p.NP * p.row_len, OVL = (long long)p.NP * p.orow_len;
  // dword loads / stores: every plane row must start on a 4-byte boundary, both sides
  if (p.row_len % 4 != 0 || p.orow_len % 4 != 0 || p.row_len < 16) return ATTWARP_OK;
  if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 3u) != 0) return ATTWARP_OK;
  if (layout == ATTWARP_HWC ? ((long long)H * W * C) % 4 != 0 || ((long long)Ho * Wo * C) % 4 != 0
                            : ((long long)H * W) % 4 != 0 || ((long long)Ho * Wo) % 4 != 0)
    return ATTWARP_OK;
  if (VL > 4096 || OVL > 4096) return ATTWARP_OK;           // 16-bit LDS offsets
  if ((long long)p.NP * ((p.row_len + 15) / 16) > u8k::NT) return ATTWARP_OK;   // one 16-byte piece per thread
  p.VL = (int)VL;
  p.OVL = (int)OVL;
  p.plane_stride = (layout == ATTWARP_HWC) ? 0 : (long long)H * W;
  p.oplane_stride = (layout == ATTWARP_HWC) ? 0 : (long long)Ho * Wo;
  p.img_stride = (long long)H * W * C;
  p.oimg_stride = (long long)Ho * Wo * C;
  if (p.plane_stride * p.NP > 2147483647LL || p.oplane_stride * p.NP > 2147483647LL) return ATTWARP_OK;
  int R = 16;
  if (const char* renv = getenv("ATTWARP_REMAP_ROWS")) { int v = atoi(renv); if (v >= 1 && v <= u8k::RMAX) R = v; }
  if (R > Ho) R = Ho;
  p.R = R;
  p.nblk = (Ho + R - 1) / R;
  const long long nb = (long long)p.nblk * B;
  if (nb > 2147483647LL) return ATTWARP_OK;
  p.nblocks = (int)nb;
  *handled = true;
  const int ko = (p.OVL + u8k::NT - 1) / u8k::NT;
  if (ko <= 4) return u8k::launch_ko<4>(p, st);
  if (ko <= 8) return u8k::launch_ko<8>(p, st);
  if (ko <= 12) return u8k::launch_ko<12>(p, st);
  return u8k::launch_ko<16>(p, st);
}

}  // namespace attwarp
