// One step of the main_batched chain (AGW/main_batched.py:243-287: revise_mask -> x255 uint8 -> PIL LANCZOS -> float64
// marginals -> cumsum CDF -> np.interp -> uint8 cv2.remap) for a STREAM of equally shaped batches, as ONE launch.
//
// The five stages of the chain depend on each other only through a batch's own intermediates, so in the steady state of
// a stream they can belong to five different batches:
//     R(k)    integer cv2 resample of batch k             remap_rows_u8i_block           HBM + LDS gathers + issue
//     F(k+1)  profile -> CDF -> inverse maps of batch k+1  attention_maps_finalize_block  a latency chain (one lane's cumsum)
//     P(k+2)  float64 marginals of the mask of batch k+2   profiles_u8_block              VALU (float64 adds in numpy's orders)
//     L(k+3)  x255 + LANCZOS up-sampling, batch k+3        lanczos_strip_block            VALU (7 integer MADs per pixel)
//     V(k+4)  revise_mask of batch k+4                     mask_postproc_block            a latency chain (576 numbers)
// Launched one behind the other (what pipeline.warp_from_masks does for one batch) every boundary costs a drain and a
// dispatch ramp and nothing overlaps although the stages are bound by different units; as branches of a HIP graph the
// hardware overlaps them but pays five launches and a fork / join per step.  Here they are block ranges of one grid:
// the F and V blocks first (the longest dependent chains start at once), then P, L and R blocks INTERLEAVED in
// proportion to their counts -- the VALU-bound and the HBM-bound blocks are resident on every CU side by side for the
// whole launch instead of one kind after the other -- in units of 8 consecutive blocks, so that block % 8 keeps naming
// the XCD for the resample's XCD-aware order.  Every body is the stand-alone kernel's body (same arithmetic, bit for
// bit); LDS is one pool sized for the largest of them.
#include "common.hpp"
#include "chain_order.hpp"
#include "mask_blocks.hpp"
#include "profiles_blocks.hpp"
#include "remap_u8_block.hpp"

#include <algorithm>
#include <cstring>

namespace attwarp {

constexpr int CHAIN_NT = 256;
constexpr int CHAIN_PRIO_DEFAULT = 0;
constexpr int CHAIN_WAVES_DEFAULT = 8;      // waves per SIMD the register allocation leaves room for

struct ChainStepArgs {
  int B;
  int prio;                                 // 1: the F and V blocks (one lane's dependent chain each) raise their wave priority
  ChainOrder ord;                           // which block does what (chain_order.hpp)
  // V: masks [B,g,g] -> rev_out
  const float* masks; int g, ks; float coe; float* rev_out;
  // L: la.mf (the rev of batch k+3) -> la.out (mota)
  LanczosStripArgs la; int l_bx;            // l_bx = nstrips * nchunks blocks per image
  // P: mota_in [B,H,W] -> col_out, ls_out; p_lut: null = identity / square (p_square) in registers, else the 256-entry
  // table of the transformed byte values (sqrt / exp / log: attwarp_attention_transform_lut)
  const uint8_t* mota_in; double* col_out; double* ls_out; const double* p_lut; int p_square;
  // F: fa.col / fa.ls -> fa.map_x / fa.map_y
  MapsFinalizeArgs fa;
  // R
  u8k::Params rp;
#ifdef ATTWARP_TUNING
  unsigned long long* trace;                // block timeline (common.hpp: trace_buffer), null = off
  int bound;                                // TUNE_BOUND (upper-bound experiments, garbage output)
#endif
};

// one block of the step; returns the kind of work it did (CHAIN_F .. CHAIN_R, CHAIN_PAD)
// XT ("extended transforms"): false = the identity transform only -- what both reference drivers pass; the kernel then is the
// one without any transform code -- true = identity / square in registers or a table (sqrt / exp / log), block uniform
template <int KI, int KD, int PD, bool XT>
__device__ __forceinline__ int chain_step_block(const ChainStepArgs& a, const PairwisePlan& Pw, const PairwisePlan& Ph, uint8_t* pool) {
  int j;
  const int kind = chain_order_decode(a.ord, blockIdx.x, j);
  if (kind == CHAIN_F) {
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    attention_maps_finalize_block(Pw, Ph, maps_finalize_image(a.fa, j >> 1, Pw.nleaves), j & 1, reinterpret_cast<double*>(pool));
  } else if (kind == CHAIN_V) {
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    float* x = reinterpret_cast<float*>(pool);
    double* red = reinterpret_cast<double*>(pool + 32 * 32 * sizeof(float));
    float* fred = reinterpret_cast<float*>(red + CHAIN_NT / WAVE);
    mask_postproc_block(a.masks, a.g, a.ks, a.coe, a.rev_out, j, x, red, fred);
  } else if (kind == CHAIN_P) {
    const int b = j / Pw.nleaves, leaf = j - b * Pw.nleaves;
#ifdef ATTWARP_TUNING
    const int bsrc = (a.bound & 2) ? (b & 1) : b;     // bound: every image reads the mask of image 0 / 1 (L2 hits)
#else
    const int bsrc = b;
#endif
    const uint8_t* att = a.mota_in + (size_t)bsrc * a.fa.h * a.fa.w;
    double* col = a.col_out + (size_t)b * a.fa.w;
    double* ls = a.ls_out + (size_t)b * a.fa.h * Pw.nleaves;
    // (block uniform: the transform of new_method.py:134-179 the caller selected; identity is what both drivers pass)
    if (XT && a.p_lut)
      profiles_u8_block<ATTWARP_T_LUT>(att, a.fa.h, a.fa.w, XfAttention<ATTWARP_T_LUT>{1.0, 1.0}, Pw.off[leaf], Pw.len[leaf], Pw.nleaves,
                                       leaf, col, ls, pool, a.p_lut);
    else if (XT && a.p_square)
      profiles_u8_block<ATTWARP_T_SQUARE>(att, a.fa.h, a.fa.w, XfAttention<ATTWARP_T_SQUARE>{1.0, 1.0}, Pw.off[leaf], Pw.len[leaf],
                                          Pw.nleaves, leaf, col, ls, pool);
    else
      profiles_u8_block<ATTWARP_T_IDENTITY>(att, a.fa.h, a.fa.w, XfAttention<ATTWARP_T_IDENTITY>{1.0, 1.0}, Pw.off[leaf], Pw.len[leaf],
                                            Pw.nleaves, leaf, col, ls, pool);
  } else if (kind == CHAIN_L) {
    const int b = j / a.l_bx, bx = j - b * a.l_bx;
#ifdef ATTWARP_TUNING
    const int bdst = (a.bound & 2) ? (b & 1) : b;     // bound: every image's mask is written over image 0 / 1
#else
    const int bdst = b;
#endif
    lanczos_strip_block<8>(a.la, bx, (size_t)b * a.la.h * a.la.w, a.la.out + (size_t)bdst * a.la.out_h * a.la.out_w, pool);
  } else if (kind == CHAIN_R) {
    u8k::remap_rows_u8i_block<KI, KD, true, PD>(a.rp, j, reinterpret_cast<float*>(pool));
  }
  return kind;
}

template <int KI, int KD, int PD, int MINW, bool XT>
__global__ __launch_bounds__(CHAIN_NT, MINW) void mask_chain_step_kernel(const ChainStepArgs a, const PairwisePlan Pw,
                                                                         const PairwisePlan Ph) {
  extern __shared__ __attribute__((aligned(16))) uint8_t pool[];
#ifdef ATTWARP_TUNING
  const TraceStart t0 = trace_now();
  const int kind = chain_step_block<KI, KD, PD, XT>(a, Pw, Ph, pool);
  trace_block(a.trace, t0, kind);
#else
  chain_step_block<KI, KD, PD, XT>(a, Pw, Ph, pool);
#endif
}

template <int KI, int KD>
static int launch_chain_kikd(const ChainStepArgs& a, const PairwisePlan& Pw, const PairwisePlan& Ph, size_t lds, unsigned grid,
                             hipStream_t st) {
  constexpr int PD = KI <= 2 ? 4 : 2;       // as launch_u8i_depth (remap_u8.hip)
#ifdef ATTWARP_TUNING
  if (tune(TUNE_CHAIN_WAVES) == 14 - CHAIN_WAVES_DEFAULT) {     // the other of {6, 8}
    hipLaunchKernelGGL((mask_chain_step_kernel<KI, KD, PD, 14 - CHAIN_WAVES_DEFAULT, false>), dim3(grid), dim3(CHAIN_NT), lds, st, a, Pw, Ph);
    return check_launch("mask_chain_step_kernel");
  }
#endif
  if (a.p_lut || a.p_square || a.fa.apply_inverse)
    hipLaunchKernelGGL((mask_chain_step_kernel<KI, KD, PD, CHAIN_WAVES_DEFAULT, true>), dim3(grid), dim3(CHAIN_NT), lds, st, a, Pw, Ph);
  else
    hipLaunchKernelGGL((mask_chain_step_kernel<KI, KD, PD, CHAIN_WAVES_DEFAULT, false>), dim3(grid), dim3(CHAIN_NT), lds, st, a, Pw, Ph);
  return check_launch("mask_chain_step_kernel");
}
template <int KI>
static int launch_chain_ki(const ChainStepArgs& a, const PairwisePlan& Pw, const PairwisePlan& Ph, size_t lds, unsigned grid,
                           int kd, hipStream_t st) {
  if (kd <= 1) return launch_chain_kikd<KI, 1>(a, Pw, Ph, lds, grid, st);
  if (kd == 2) return launch_chain_kikd<KI, 2>(a, Pw, Ph, lds, grid, st);
  if (kd == 3) return launch_chain_kikd<KI, 3>(a, Pw, Ph, lds, grid, st);
  return launch_chain_kikd<KI, 4>(a, Pw, Ph, lds, grid, st);
}

}  // namespace attwarp

using namespace attwarp;

extern "C" int attwarp_mask_chain_step(const uint8_t* images, uint8_t* out, int B, int C, int H, int W, int H_out, int W_out,
                                       const float* map_x, const float* map_y,
                                       const void* sums_in, float* map_x_next, float* map_y_next,
                                       const uint8_t* mota_in, void* sums_out,
                                       const float* rev_in, const int32_t* bounds_x, const int32_t* kk_x, int ksize_x,
                                       const int32_t* bounds_y, const int32_t* kk_y, int ksize_y, uint8_t* mota_out,
                                       const float* masks, int g, int kernel_size, float enhance_coe, float* rev_out,
                                       int transform, double exp_scale, double exp_divisor, int apply_inverse,
                                       const double* transform_lut, void* stream) {
  ATTWARP_REQUIRE(images && out && map_x && map_y, "mask_chain_step: null image / map pointer");
  ATTWARP_REQUIRE(transform >= ATTWARP_T_IDENTITY && transform <= ATTWARP_T_LOG, "mask_chain_step: unknown transform %d", transform);
  ATTWARP_REQUIRE(transform <= ATTWARP_T_SQUARE || transform_lut,
                  "mask_chain_step: the sqrt / exp / log transforms need transform_lut (attwarp_attention_transform_lut)");
  ATTWARP_REQUIRE(sums_in && map_x_next && map_y_next && mota_in && sums_out && rev_in && bounds_x && kk_x && bounds_y && kk_y &&
                  mota_out && masks && rev_out, "mask_chain_step: null stage pointer");
  ATTWARP_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && H_out > 0 && W_out > 0 && g > 0, "mask_chain_step: non-positive size");
  ATTWARP_REQUIRE(kernel_size > 0 && (kernel_size & 1), "mask_chain_step: kernel_size must be odd (got %d)", kernel_size);
  ATTWARP_REQUIRE(map_x_next != map_x && map_y_next != map_y, "mask_chain_step: the next maps must not alias the current ones");
  ATTWARP_REQUIRE(sums_out != sums_in && mota_out != mota_in && rev_out != rev_in,
                  "mask_chain_step: a stage's output buffer must not alias the buffer the next stage reads in the same launch");
  if (g > 32 || kernel_size > 7 || B > 65535) return fail(ATTWARP_E_UNSUPPORTED, "mask_chain_step: g <= 32, kernel_size <= 7, B <= 65535");
  // L: the column-strip kernel's conditions (attwarp_mask_upsample_lanczos)
  if (W == g || H == g || W % 4 != 0 || ksize_x > 8 || ksize_y != 8 || (reinterpret_cast<uintptr_t>(mota_out) & 3u) != 0 ||
      lanczos_strip_lds_bytes(g, g) > 48 * 1024)
    return fail(ATTWARP_E_UNSUPPORTED, "mask_chain_step: the mask up-sampling of this shape does not run on the column-strip kernel");
  // P: the byte-packed profile kernel's conditions (launch_profiles_u8)
  PairwisePlan Pw, Ph;
  if (!pw_build(W, Pw) || !pw_build(H, Ph)) return fail(ATTWARP_E_UNSUPPORTED, "mask_chain_step: max(H,W) > 16384");
  if ((reinterpret_cast<uintptr_t>(mota_in) & 3u) != 0)
    return fail(ATTWARP_E_UNSUPPORTED, "mask_chain_step: the mask buffers must be 4-byte aligned");
  for (int j = 0; j < Pw.nleaves; ++j)
    if (Pw.len[j] < 8 || Pw.len[j] % 4 != 0) return fail(ATTWARP_E_UNSUPPORTED, "mask_chain_step: W=%d does not split into leaves of multiples of 4", W);
  ChainStepArgs a;
  memset(&a, 0, sizeof(a));
  // R: the integer cv2 kernel's conditions
  if (!u8i_params(a.rp, images, out, ATTWARP_HWC, B, C, H, W, H_out, W_out, map_x, map_y) || a.rp.unaligned)
    return fail(ATTWARP_E_UNSUPPORTED, "mask_chain_step: this image shape / alignment does not run on the integer cv2 resample");
  a.B = B;
  a.prio = tune(TUNE_STEP_PRIO) >= 0 ? tune(TUNE_STEP_PRIO) : CHAIN_PRIO_DEFAULT;
  a.masks = masks; a.g = g; a.ks = kernel_size; a.coe = enhance_coe; a.rev_out = rev_out;
  {   // L geometry: as attwarp_mask_upsample_lanczos
    const int nstrips = (W + MASK_NT - 1) / MASK_NT;
    long long nchunks = (4096 + (long long)B * nstrips - 1) / ((long long)B * nstrips);
    const int max_chunks = (H + 63) / 64;
    if (nchunks > max_chunks) nchunks = max_chunks;
    if (nchunks < 1) nchunks = 1;
    const int rows_per_chunk = (int)((H + nchunks - 1) / nchunks);
    nchunks = (H + rows_per_chunk - 1) / rows_per_chunk;
    a.la = LanczosStripArgs{rev_in, nullptr, g, g, H, W, bounds_x, kk_x, ksize_x, bounds_y, kk_y, (int)nchunks, rows_per_chunk, mota_out};
    a.l_bx = nstrips * (int)nchunks;
  }
  a.mota_in = mota_in;
  a.col_out = static_cast<double*>(sums_out);
  a.ls_out = a.col_out + (size_t)B * W;
  const double* col_in = static_cast<const double*>(sums_in);
  a.p_lut = transform > ATTWARP_T_SQUARE ? transform_lut : nullptr;
  a.p_square = transform == ATTWARP_T_SQUARE;
  a.fa = MapsFinalizeArgs{col_in, col_in + (size_t)B * W, H, W, W_out, H_out, transform, exp_scale, exp_divisor, apply_inverse ? 1 : 0,
                          map_x_next, map_y_next, pw_depth(Pw)};
  a.ord.nF = 2 * B; a.ord.nV = B;
  a.ord.nF8 = (2 * B + 7) / 8;
  a.ord.nV8 = (B + 7) / 8;
  a.ord.nP = Pw.nleaves * B;
  a.ord.nL = a.l_bx * B;
  a.ord.nR = a.rp.nblocks;
  build_interleave(a.ord, tune(TUNE_CHAIN_SEQ) >= 0 ? tune(TUNE_CHAIN_SEQ) : CHAIN_ORDER_DEFAULT);
#ifdef ATTWARP_TUNING
  a.trace = trace_buffer();
  a.fa.trace = a.trace;
  a.bound = tune(TUNE_BOUND) > 0 ? tune(TUNE_BOUND) : 0;
  a.fa.bound = a.bound;
#endif
  const long long octs = chain_order_octets(a.ord);
  if (octs * 8 > 2147483647LL) return fail(ATTWARP_E_UNSUPPORTED, "mask_chain_step: grid too large");
  size_t lds = std::max(std::max(u8k::u8i_lds_bytes(), a.p_lut ? profiles_u8_lds_bytes<ATTWARP_T_LUT>() : profiles_u8_lds_bytes<ATTWARP_T_IDENTITY>()),
                        std::max(maps_finalize_lds_bytes(H, W, Pw, Ph), std::max(lanczos_strip_lds_bytes(g, g), mask_postproc_lds_bytes())));
  if (lds > LDS_DEFAULT_MAX) return fail(ATTWARP_E_UNSUPPORTED, "mask_chain_step: %zu bytes of LDS per workgroup (> %zu)", lds, LDS_DEFAULT_MAX);
  const int ki = ((a.rp.VL >> 2) + u8k::NT - 1) / u8k::NT, kd = ((a.rp.OVL >> 2) + u8k::NT - 1) / u8k::NT;
  hipStream_t st = as_stream(stream);
  const unsigned grid = (unsigned)(octs * 8);
  switch (ki) {
    case 1: return launch_chain_ki<1>(a, Pw, Ph, lds, grid, kd, st);
    case 2: return launch_chain_ki<2>(a, Pw, Ph, lds, grid, kd, st);
    case 3: return launch_chain_ki<3>(a, Pw, Ph, lds, grid, kd, st);
    default: return launch_chain_ki<4>(a, Pw, Ph, lds, grid, kd, st);
  }
}
