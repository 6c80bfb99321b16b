// The main_batched chain (AGW/main_batched.py:243-287: revise_mask -> x255 uint8 -> PIL LANCZOS to the image's size ->
// float64 marginals -> cumsum CDF -> np.interp -> uint8 cv2.remap to 500 x 500) for batches of DIFFERENTLY sized images.
//
// The reference's driver holds a batch of PIL images at their native sizes (`b_images[j]`, main_batched.py:246;
// blend_mask up-samples the mask to `image.size`, llava.py:253; save_warped_image warps at that size, new_method.py:
// 415-422,478-488).  chain_step.hip serves batches of ONE shape; here every image of a batch has its own (H, W):
//
//   * a batch is described by a TABLE (attwarp_ragged_plan, host code below): per image its pointer, size, Pillow
//     coefficient tables and the offsets of its intermediates in the batch's packed buffers; numpy's pairwise-summation
//     plans of the distinct widths / heights; and for the two stages whose block count depends on the image (L, P) a
//     block -> (image, sub-block) map.  The table is position independent: the caller copies it to the device as is.
//   * the kernel is the stream step of chain_step.hip with every per-image quantity read from the table: R(k) | F(k+1) |
//     P(k+2) | L(k+3) | V(k+4) as block ranges of ONE launch, the same bodies (remap_rows_u8i_rows, attention_maps_
//     finalize_block, profiles_u8_block, lanczos_strip_block, mask_postproc_block: bit for bit what the stand-alone entry
//     points compute), in their "unaligned" forms: 683-pixel-wide images have rows of 2049 bytes.  Any stage may be absent
//     (null table), so ONE batch is five launches of the same kernel (pipeline.warp_from_masks_ragged) and a stream of
//     batches is one launch per batch (pipeline.RaggedMaskChainStream).
//   * output is dense: [B, H_out, W_out, C] whatever the input sizes, like the reference's 500 x 500 PNGs.
#include "common.hpp"
#include "chain_order.hpp"
#include "mask_blocks.hpp"
#include "profiles_blocks.hpp"
#include "remap_u8_block.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace attwarp {

constexpr int RAGGED_NT = 256;
constexpr int RAGGED_WAVES = 8;             // waves per SIMD the register allocation leaves room for (as chain_step.hip)
constexpr uint32_t RAGGED_MAGIC = 0x52474131u;   // "RGA1"
constexpr int RAGGED_MAX_LEAVES = ATTWARP_RAGGED_MAX_LEAVES;

// numpy's pairwise plan of one length (profiles_blocks.hpp: PairwisePlan) in the table: same member names, so the
// bodies' plan_to_lds / pw_combine read it from global memory as they read a PairwisePlan from the kernel arguments
struct RaggedPlan {
  int32_t nleaves, nprog, depth, n;
  int32_t off[RAGGED_MAX_LEAVES];
  int32_t len[RAGGED_MAX_LEAVES];
  unsigned char prog[2 * RAGGED_MAX_LEAVES];
};

struct RaggedImage {                      // one image of a batch, as the kernel reads it
  const uint8_t* image;
  const int32_t* bounds_x; const int32_t* kk_x; const int32_t* bounds_y; const int32_t* kk_y;
  int32_t H, W, ksize_x, plan_w;          // plan_w / plan_h: index into the table's plans
  int32_t plan_h, l_nchunks, l_rows_per_chunk, ki;   // ki: source dwords per thread of the resample (1..4)
  int64_t mota_off;                       // bytes: this image's up-sampled mask [H,W] in the batch's mask buffer
  int64_t sums_off;                       // doubles: col[W] | ls[H, nleaves(W)] in the batch's axis-sum workspace
};
static_assert(sizeof(RaggedImage) == 88, "table layout");
static_assert(sizeof(attwarp_ragged_header) % 8 == 0, "table layout");

struct RaggedStepArgs {
  ChainOrder ord;
  int prio;
  int g, ks, C, Ho, Wo;
  float coe;
  // V(k+4)
  const float* masks; float* rev_out;
  // L(k+3)
  const RaggedImage* l_img; const uint32_t* l_map; const float* rev_in; uint8_t* mota_out;
  // P(k+2): p_lut null = identity / square (p_square) in registers, else the table of the transformed byte values (sqrt / exp / log)
  const RaggedImage* p_img; const RaggedPlan* p_plans; const uint32_t* p_map; const uint8_t* mota_in; double* sums_out;
  const double* p_lut; int p_square;
  // F(k+1)
  const RaggedImage* f_img; const RaggedPlan* f_plans; const uint32_t* f_order; const double* sums_in; float* map_x_next; float* map_y_next;
  int transform, apply_inverse; double exp_scale, exp_divisor;
  // R(k)
  const RaggedImage* r_img; const uint32_t* r_order; uint8_t* out; const float* map_x; const float* map_y;
  int r_rows, r_nblk;                       // output rows per resample block, blocks per image
#ifdef ATTWARP_TUNING
  unsigned long long* trace;
#endif
};

// one block of the step; returns the kind of work it did (CHAIN_F .. CHAIN_R, CHAIN_PAD)
// XT ("extended transforms"): false = the identity transform only (what both reference drivers pass: the kernel without any
// transform code), true = identity / square in registers or a table (sqrt / exp / log), block uniform
template <int KD, bool XT>
__device__ __forceinline__ int ragged_step_block(const RaggedStepArgs& a, uint8_t* pool) {
  int j;
  const int kind = chain_order_decode(a.ord, blockIdx.x, j);
  if (kind == CHAIN_F) {
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    const int b = (int)a.f_order[j >> 1];           // largest images first: their chains are the longest blocks of the launch
    const RaggedImage& im = a.f_img[b];
    const RaggedPlan& Pw = a.f_plans[im.plan_w];
    const RaggedPlan& Ph = a.f_plans[im.plan_h];
    const double* col = a.sums_in + im.sums_off;
    MapsFinalizeArgs fa{col, col + im.W, im.H, im.W, a.Wo, a.Ho, XT ? a.transform : (int)ATTWARP_T_IDENTITY, XT ? a.exp_scale : 1.0,
                        XT ? a.exp_divisor : 1.0, XT ? a.apply_inverse : 0,
                        a.map_x_next + (size_t)b * a.Wo, a.map_y_next + (size_t)b * a.Ho, Pw.depth};
#ifdef ATTWARP_TUNING
    fa.trace = a.trace;
#endif
    attention_maps_finalize_block(Pw, Ph, fa, j & 1, reinterpret_cast<double*>(pool));
  } else if (kind == CHAIN_V) {
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    float* x = reinterpret_cast<float*>(pool);
    double* red = reinterpret_cast<double*>(pool + 32 * 32 * sizeof(float));
    float* fred = reinterpret_cast<float*>(red + RAGGED_NT / WAVE);
    mask_postproc_block(a.masks, a.g, a.ks, a.coe, a.rev_out, j, x, red, fred);
  } else if (kind == CHAIN_P) {
    const uint32_t e = a.p_map[j];
    const int b = (int)(e & 0xffffu), leaf = (int)(e >> 16);
    const RaggedImage& im = a.p_img[b];
    const RaggedPlan& Pw = a.p_plans[im.plan_w];
    double* col = a.sums_out + im.sums_off;
    const uint8_t* att = a.mota_in + im.mota_off;
    // (block uniform: the transform of new_method.py:134-179 the caller selected; identity is what both drivers pass)
    if (XT && a.p_lut)
      profiles_u8_block<ATTWARP_T_LUT, true>(att, im.H, im.W, XfAttention<ATTWARP_T_LUT>{1.0, 1.0}, Pw.off[leaf], Pw.len[leaf], Pw.nleaves,
                                             leaf, col, col + im.W, pool, a.p_lut);
    else if (XT && a.p_square)
      profiles_u8_block<ATTWARP_T_SQUARE, true>(att, im.H, im.W, XfAttention<ATTWARP_T_SQUARE>{1.0, 1.0}, Pw.off[leaf], Pw.len[leaf],
                                                Pw.nleaves, leaf, col, col + im.W, pool);
    else
      profiles_u8_block<ATTWARP_T_IDENTITY, true>(att, im.H, im.W, XfAttention<ATTWARP_T_IDENTITY>{1.0, 1.0}, Pw.off[leaf], Pw.len[leaf],
                                                  Pw.nleaves, leaf, col, col + im.W, pool);
  } else if (kind == CHAIN_L) {
    const uint32_t e = a.l_map[j];
    const int b = (int)(e & 0xffffu), bx = (int)(e >> 16);
    const RaggedImage& im = a.l_img[b];
    const LanczosStripArgs la{a.rev_in, nullptr, a.g, a.g, im.H, im.W, im.bounds_x, im.kk_x, im.ksize_x, im.bounds_y, im.kk_y,
                              im.l_nchunks, im.l_rows_per_chunk, nullptr};
    lanczos_strip_block<8, true>(la, bx, (size_t)b * a.g * a.g, a.mota_out + im.mota_off, pool);
  } else if (kind == CHAIN_R) {
    // Whole images per XCD (block % 8 names the XCD: the halo row two neighbouring row blocks share is an L2 hit), dealt
    // round robin from the size-sorted order so that every XCD gets large and small images alike: image rank s * 8 + xcd.
    // (A batch that is not a multiple of 8 images takes the plain order: consecutive row blocks on consecutive XCDs.)
    int bi, rb0;
    if ((a.ord.nR / a.r_nblk) % 8 == 0) {
      const int xcd = j & 7, idx = j >> 3, sl = idx / a.r_nblk;
      bi = sl * 8 + xcd;
      rb0 = idx - sl * a.r_nblk;
    } else {
      bi = j / a.r_nblk;
      rb0 = j - bi * a.r_nblk;
    }
    const int b = (int)a.r_order[bi];
    const RaggedImage& im = a.r_img[b];
    u8k::Params p;
    p.H = im.H; p.W = im.W; p.Ho = a.Ho; p.Wo = a.Wo;
    p.NP = 1; p.CS = a.C;
    p.row_len = p.VL = im.W * a.C;
    p.orow_len = p.OVL = a.Wo * a.C;
    p.plane_stride = p.oplane_stride = 0;
    p.R = a.r_rows; p.nblk = p.wpi = a.r_nblk;
    const int img_bytes = im.H * p.row_len, oimg_bytes = a.Ho * p.orow_len;
    uint8_t* dst_b = a.out + (size_t)b * oimg_bytes;
    const float* mx_b = a.map_x + (size_t)b * a.Wo;
    const float* my_b = a.map_y + (size_t)b * a.Ho;
    float* smem = reinterpret_cast<float*>(pool);
    // block uniform: source dwords per thread (rows requested ahead: as launch_u8i_depth, remap_u8.hip) x whether THIS image's
    // rows are dword aligned -- the unaligned form holds two aligned dwords per row dword until the row is consumed, so it
    // requests its rows half as far ahead to stay inside the kernel's register budget
    // (an output row that is not a multiple of 4 bytes -- or an output image that does not start on a dword -- needs the
    //  unaligned form too: only it stores a row's last 1..3 bytes)
    const bool ua = ((p.row_len | p.orow_len | (int)(reinterpret_cast<uintptr_t>(im.image) & 3u) |
                      (int)(reinterpret_cast<uintptr_t>(dst_b) & 3u)) & 3) != 0;
#define ATTWARP_RAGGED_R(KI_, PD_, UA_) \
    u8k::remap_rows_u8i_rows<KI_, KD, true, PD_, UA_>(p, im.image, img_bytes, dst_b, oimg_bytes, mx_b, my_b, rb0, smem)
    if (!ua) {
      switch (im.ki) {
        case 1: ATTWARP_RAGGED_R(1, 4, false); break;
        case 2: ATTWARP_RAGGED_R(2, 4, false); break;
        case 3: ATTWARP_RAGGED_R(3, 2, false); break;
        default: ATTWARP_RAGGED_R(4, 2, false); break;
      }
    } else {
      switch (im.ki) {
        case 1: ATTWARP_RAGGED_R(1, 4, true); break;
        case 2: ATTWARP_RAGGED_R(2, 2, true); break;
        case 3: ATTWARP_RAGGED_R(3, 1, true); break;
        default: ATTWARP_RAGGED_R(4, 1, true); break;
      }
    }
#undef ATTWARP_RAGGED_R
  }
  return kind;
}

template <int KD, bool XT>
__global__ __launch_bounds__(RAGGED_NT, RAGGED_WAVES) void mask_chain_ragged_kernel(const RaggedStepArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t pool[];
#ifdef ATTWARP_TUNING
  const TraceStart t0 = trace_now();
  const int kind = ragged_step_block<KD, XT>(a, pool);
  trace_block(a.trace, t0, kind);
#else
  ragged_step_block<KD, XT>(a, pool);
#endif
}

// ---- host: Pillow's coefficient tables (ImagingResample precompute_coeffs + normalize_coeffs_8bpc, Resample.c) -------
static double pil_sinc(double x) {
  if (x == 0.0) return 1.0;
  x *= M_PI;
  return sin(x) / x;
}
static double pil_lanczos3(double x) { return (-3.0 <= x && x < 3.0) ? pil_sinc(x) * pil_sinc(x / 3.0) : 0.0; }
static double pil_bicubic(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

// ---- host: the table of one batch ----------------------------------------------------------------------------------
struct RaggedLayout {
  attwarp_ragged_header h;
  std::vector<RaggedImage> img;
  std::vector<RaggedPlan> plans;
  std::vector<uint32_t> lmap, pmap, order;
};

static bool ragged_plan_of(int n, RaggedPlan& out) {
  PairwisePlan P;
  if (!pw_build(n, P) || P.nleaves > RAGGED_MAX_LEAVES) return false;
  memset(&out, 0, sizeof(out));
  out.nleaves = P.nleaves; out.nprog = P.nprog; out.depth = pw_depth(P); out.n = n;
  for (int i = 0; i < P.nleaves; ++i) { out.off[i] = P.off[i]; out.len[i] = P.len[i]; }
  for (int i = 0; i < P.nprog; ++i) out.prog[i] = P.prog[i];
  return true;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// fills L from the caller's images; returns 0 or an error code (message set)
static int ragged_layout(const attwarp_ragged_image* images, int B, int C, int g, int Ho, int Wo, RaggedLayout& L) {
  ATTWARP_REQUIRE(images, "ragged_plan: null image table");
  ATTWARP_REQUIRE(B > 0 && C > 0 && g > 0 && Ho > 0 && Wo > 0, "ragged_plan: non-positive size");
  if (B > 65535 || C > 4 || g > 32) return fail(ATTWARP_E_UNSUPPORTED, "ragged_plan: B <= 65535, C <= 4, g <= 32");
  const long long OVL = (long long)Wo * C;
  if (OVL > 4096 || (long long)Ho * OVL > 2147483647LL || Ho > 65535)
    return fail(ATTWARP_E_UNSUPPORTED, "ragged_plan: output rows wider than 4096 bytes do not run on the integer cv2 resample");
  if (lanczos_strip_lds_bytes(g, g) > 48 * 1024) return fail(ATTWARP_E_UNSUPPORTED, "ragged_plan: g too large for the column-strip up-sampling");
  memset(&L.h, 0, sizeof(L.h));
  L.img.resize(B);
  size_t mota = 0, sums = 0, lds = 0;
  int max_hw = 0;
  // enough up-sampling blocks to fill the chip (as attwarp_mask_upsample_lanczos: ~4096 per launch, >= 64 rows per chunk)
  const long long per_image = (4096 + B - 1) / B;
  for (int b = 0; b < B; ++b) {
    const attwarp_ragged_image& in = images[b];
    ATTWARP_REQUIRE(in.image && in.bounds_x && in.kk_x && in.bounds_y && in.kk_y, "ragged_plan: null pointer in image %d", b);
    ATTWARP_REQUIRE(in.H > 0 && in.W > 0 && in.ksize_x > 0, "ragged_plan: non-positive size in image %d", b);
    const long long VL = (long long)in.W * C;
    if (VL > 4096 || VL < 4 || (long long)in.H * VL > 2147483647LL)
      return fail(ATTWARP_E_UNSUPPORTED, "ragged_plan: image %d (%d x %d x %d): rows of 4 .. 4096 bytes run on the integer cv2 resample", b, in.H, in.W, C);
    if (in.H <= g || in.W <= g || in.ksize_x > 8)
      return fail(ATTWARP_E_UNSUPPORTED, "ragged_plan: image %d (%d x %d): the mask must be up-sampled on both axes (<= 8 taps)", b, in.H, in.W);
    RaggedImage& im = L.img[b];
    memset(&im, 0, sizeof(im));
    im.image = in.image; im.bounds_x = in.bounds_x; im.kk_x = in.kk_x; im.bounds_y = in.bounds_y; im.kk_y = in.kk_y;
    im.H = in.H; im.W = in.W; im.ksize_x = in.ksize_x;
    int pw = -1, ph = -1;
    for (size_t i = 0; i < L.plans.size(); ++i) {
      if (L.plans[i].n == in.W) pw = (int)i;
      if (L.plans[i].n == in.H) ph = (int)i;
    }
    for (int axis = 0; axis < 2; ++axis) {
      int& idx = axis ? ph : pw;
      const int n = axis ? in.H : in.W;
      if (idx < 0 && axis == 1 && in.H == in.W) idx = pw;
      if (idx >= 0) continue;
      RaggedPlan P;
      if (!ragged_plan_of(n, P)) return fail(ATTWARP_E_UNSUPPORTED, "ragged_plan: image %d: max(H,W) > %d", b, 128 * RAGGED_MAX_LEAVES);
      L.plans.push_back(P);
      idx = (int)L.plans.size() - 1;
    }
    im.plan_w = pw; im.plan_h = ph;
    const RaggedPlan& Pw = L.plans[pw];
    for (int jl = 0; jl < Pw.nleaves; ++jl)
      if (Pw.len[jl] < 8) return fail(ATTWARP_E_UNSUPPORTED, "ragged_plan: image %d: W=%d has a leaf shorter than 8", b, in.W);
    // L: column strips of 256 pixels x chunks of rows
    const int nstrips = (in.W + MASK_NT - 1) / MASK_NT;
    long long nchunks = (per_image + nstrips - 1) / nstrips;
    const int max_chunks = (in.H + 63) / 64;
    if (nchunks > max_chunks) nchunks = max_chunks;
    if (nchunks < 1) nchunks = 1;
    const int rows_per_chunk = (int)((in.H + nchunks - 1) / nchunks);
    nchunks = (in.H + rows_per_chunk - 1) / rows_per_chunk;
    im.l_nchunks = (int)nchunks; im.l_rows_per_chunk = rows_per_chunk;
    if (nstrips * nchunks > 65535) return fail(ATTWARP_E_UNSUPPORTED, "ragged_plan: image %d: too many up-sampling blocks", b);
    im.ki = (int)((((VL + 3) >> 2) + u8k::NT - 1) / u8k::NT);
    im.mota_off = (int64_t)mota;
    mota = align_up(mota + (size_t)in.H * in.W, 256);
    im.sums_off = (int64_t)sums;
    sums += (size_t)in.W + (size_t)in.H * Pw.nleaves;
    max_hw = std::max(max_hw, std::max(in.H, in.W));
    // F's LDS (maps_finalize_lds_bytes) for this image
    const RaggedPlan& Ph = L.plans[ph];
    const int nl = std::max(Pw.nleaves, Ph.nleaves);
    const size_t f = (size_t)(std::max(in.H, in.W) + 2) * 8 + (PROF_NT / WAVE) * 8 + (size_t)Pw.depth * PROF_NT * 8 + (size_t)nl * 8 +
                     plan_view_lds_bytes(Pw.nleaves) + plan_view_lds_bytes(Ph.nleaves);
    lds = std::max(lds, f);
  }
  // Block order inside every stage: the largest images first.  A block's duration grows with its image (the marginals
  // blocks walk all H rows of a 128-column strip, the finalize blocks run an H- / W-long dependent chain): dispatched in
  // batch order a large image near the end of the batch would be the tail of the launch.
  L.order.resize(B);
  for (int b = 0; b < B; ++b) L.order[b] = (uint32_t)b;
  std::stable_sort(L.order.begin(), L.order.end(), [&](uint32_t x, uint32_t y) {
    return (long long)L.img[x].H * L.img[x].W > (long long)L.img[y].H * L.img[y].W;
  });
  for (uint32_t b : L.order) {
    const RaggedImage& im = L.img[b];
    const int nstrips = (im.W + MASK_NT - 1) / MASK_NT;
    for (int sblk = 0; sblk < nstrips * im.l_nchunks; ++sblk) L.lmap.push_back(b | ((uint32_t)sblk << 16));
    for (int jl = 0; jl < L.plans[im.plan_w].nleaves; ++jl) L.pmap.push_back(b | ((uint32_t)jl << 16));
  }
  lds = std::max(lds, std::max(std::max(u8k::u8i_lds_bytes(), profiles_u8_lds_bytes<ATTWARP_T_IDENTITY>()),
                               std::max(lanczos_strip_lds_bytes(g, g), mask_postproc_lds_bytes())));
  if (lds > LDS_DEFAULT_MAX) return fail(ATTWARP_E_UNSUPPORTED, "ragged_plan: %zu bytes of LDS per workgroup (> %zu)", lds, LDS_DEFAULT_MAX);
  attwarp_ragged_header& h = L.h;
  h.magic = RAGGED_MAGIC; h.B = B; h.C = C; h.g = g; h.H_out = Ho; h.W_out = Wo;
  h.nL = (int32_t)L.lmap.size(); h.nP = (int32_t)L.pmap.size();
  h.rows_per_block = (OVL >= 2048) ? 32 : 16;           // as plan_u8 (remap_u8.hip)
  if (const int v = tune(TUNE_REMAP_ROWS); v >= 1 && v <= u8k::RMAX) h.rows_per_block = v;     // (tuning flavour: sweeps)
  if (h.rows_per_block > Ho) h.rows_per_block = Ho;
  h.blocks_per_image = (Ho + h.rows_per_block - 1) / h.rows_per_block;
  if ((long long)h.blocks_per_image * B > 2147483647LL) return fail(ATTWARP_E_UNSUPPORTED, "ragged_plan: grid too large");
  h.nR = h.blocks_per_image * B;
  h.nplans = (int32_t)L.plans.size();
  h.kd = (int32_t)((((OVL + 3) >> 2) + u8k::NT - 1) / u8k::NT);
  h.max_hw = max_hw;
  h.mota_bytes = mota; h.sums_bytes = sums * sizeof(double); h.lds_bytes = lds;
  h.off_images = sizeof(attwarp_ragged_header);
  h.off_plans = align_up(h.off_images + (size_t)B * sizeof(RaggedImage), 8);
  h.off_lmap = h.off_plans + L.plans.size() * sizeof(RaggedPlan);
  h.off_pmap = h.off_lmap + L.lmap.size() * sizeof(uint32_t);
  h.off_order = h.off_pmap + L.pmap.size() * sizeof(uint32_t);
  h.table_bytes = align_up(h.off_order + L.order.size() * sizeof(uint32_t), 16);
  return ATTWARP_OK;
}

static const attwarp_ragged_header* checked_header(const void* host, const char* which) {
  const attwarp_ragged_header* h = static_cast<const attwarp_ragged_header*>(host);
  if (h->magic != RAGGED_MAGIC) { fail(ATTWARP_E_ARG, "mask_chain_ragged: the %s table was not written by attwarp_ragged_plan", which); return nullptr; }
  return h;
}

template <int KD>
static int launch_ragged(const RaggedStepArgs& a, size_t lds, unsigned grid, hipStream_t st) {
  if (a.p_lut || a.p_square || a.apply_inverse)
    hipLaunchKernelGGL((mask_chain_ragged_kernel<KD, true>), dim3(grid), dim3(RAGGED_NT), lds, st, a);
  else
    hipLaunchKernelGGL((mask_chain_ragged_kernel<KD, false>), dim3(grid), dim3(RAGGED_NT), lds, st, a);
  return check_launch("mask_chain_ragged_kernel");
}

}  // namespace attwarp

using namespace attwarp;

extern "C" int attwarp_pil_coeffs_8bpc(int in_size, int out_size, int filter, int32_t* bounds, int32_t* kk, int kk_cols) try {
  ATTWARP_REQUIRE(bounds && kk, "pil_coeffs_8bpc: null pointer");
  ATTWARP_REQUIRE(in_size > 0 && out_size > 0 && kk_cols > 0, "pil_coeffs_8bpc: non-positive size");
  ATTWARP_REQUIRE(filter == ATTWARP_PIL_LANCZOS || filter == ATTWARP_PIL_BICUBIC, "pil_coeffs_8bpc: unknown filter %d", filter);
  double (*fn)(double) = filter == ATTWARP_PIL_LANCZOS ? pil_lanczos3 : pil_bicubic;
  const double support0 = filter == ATTWARP_PIL_LANCZOS ? 3.0 : 2.0;
  constexpr int PRECISION_BITS = 32 - 8 - 2;
  if (in_size == out_size) {     // Pillow skips the pass; an identity table makes clip8((v << 22 + 2^21) >> 22) == v
    for (int o = 0; o < out_size; ++o) {
      bounds[2 * o] = o; bounds[2 * o + 1] = 1;
      for (int i = 0; i < kk_cols; ++i) kk[(size_t)o * kk_cols + i] = i == 0 ? (1 << PRECISION_BITS) : 0;
    }
    return 1;
  }
  const double scale = (double)in_size / (double)out_size;
  const double fscale = scale < 1.0 ? 1.0 : scale;
  const double support = support0 * fscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  if (ksize > kk_cols) return fail(ATTWARP_E_ARG, "pil_coeffs_8bpc: %d taps do not fit %d columns", ksize, kk_cols);
  const double ss = 1.0 / fscale;
  std::vector<double> w((size_t)ksize);
  for (int o = 0; o < out_size; ++o) {
    const double center = (o + 0.5) * scale;
    int lo = (int)(center - support + 0.5);
    if (lo < 0) lo = 0;
    int hi = (int)(center + support + 0.5);
    if (hi > in_size) hi = in_size;
    const int cnt = hi - lo;
    double tot = 0.0;
    for (int i = 0; i < cnt; ++i) {
      w[i] = fn((i + lo - center + 0.5) * ss);
      tot += w[i];
    }
    for (int i = 0; i < cnt; ++i)
      if (tot != 0.0) w[i] /= tot;
    int32_t* row = kk + (size_t)o * kk_cols;
    for (int i = 0; i < kk_cols; ++i) row[i] = 0;
    for (int i = 0; i < cnt; ++i) row[i] = (int32_t)((w[i] < 0 ? -0.5 : 0.5) + w[i] * (1 << PRECISION_BITS));
    bounds[2 * o] = lo; bounds[2 * o + 1] = cnt;
  }
  return ksize;
} catch (...) {
  return fail(ATTWARP_E_UNSUPPORTED, "pil_coeffs_8bpc: out of host memory");
}

// (the host helpers allocate std::vectors: an allocation failure must not travel through the C ABI as an exception)
extern "C" size_t attwarp_ragged_table_bytes(const attwarp_ragged_image* images, int B, int C, int g, int H_out, int W_out) try {
  RaggedLayout L;
  if (ragged_layout(images, B, C, g, H_out, W_out, L) != ATTWARP_OK) return 0;
  return (size_t)L.h.table_bytes;
} catch (...) {
  fail(ATTWARP_E_UNSUPPORTED, "ragged_table_bytes: out of host memory");
  return 0;
}

extern "C" int attwarp_ragged_plan(const attwarp_ragged_image* images, int B, int C, int g, int H_out, int W_out, void* table,
                                   size_t table_bytes) try {
  ATTWARP_REQUIRE(table, "ragged_plan: null table");
  RaggedLayout L;
  if (const int rc = ragged_layout(images, B, C, g, H_out, W_out, L)) return rc;
  ATTWARP_REQUIRE(table_bytes >= L.h.table_bytes, "ragged_plan: the table needs %llu bytes (got %zu)", (unsigned long long)L.h.table_bytes, table_bytes);
  uint8_t* t = static_cast<uint8_t*>(table);
  memset(t, 0, (size_t)L.h.table_bytes);
  memcpy(t, &L.h, sizeof(L.h));
  memcpy(t + L.h.off_images, L.img.data(), L.img.size() * sizeof(RaggedImage));
  memcpy(t + L.h.off_plans, L.plans.data(), L.plans.size() * sizeof(RaggedPlan));
  memcpy(t + L.h.off_lmap, L.lmap.data(), L.lmap.size() * sizeof(uint32_t));
  memcpy(t + L.h.off_pmap, L.pmap.data(), L.pmap.size() * sizeof(uint32_t));
  memcpy(t + L.h.off_order, L.order.data(), L.order.size() * sizeof(uint32_t));
  return ATTWARP_OK;
} catch (...) {
  return fail(ATTWARP_E_UNSUPPORTED, "ragged_plan: out of host memory");
}

extern "C" int attwarp_mask_chain_ragged(const void* r_host, const void* r_dev, uint8_t* out, const float* map_x, const float* map_y,
                                         const void* f_host, const void* f_dev, const void* sums_in, float* map_x_next, float* map_y_next,
                                         const void* p_host, const void* p_dev, const uint8_t* mota_in, void* sums_out,
                                         const void* l_host, const void* l_dev, const float* rev_in, uint8_t* mota_out,
                                         const float* masks, int B_masks, int g, int kernel_size, float enhance_coe, float* rev_out,
                                         int transform, double exp_scale, double exp_divisor, int apply_inverse,
                                         const double* transform_lut, void* stream) {
  ATTWARP_REQUIRE(r_host || f_host || p_host || l_host || masks, "mask_chain_ragged: no stage given");
  ATTWARP_REQUIRE(transform >= ATTWARP_T_IDENTITY && transform <= ATTWARP_T_LOG, "mask_chain_ragged: unknown transform %d", transform);
  ATTWARP_REQUIRE(!p_host || transform <= ATTWARP_T_SQUARE || transform_lut,
                  "mask_chain_ragged: the sqrt / exp / log transforms need transform_lut (attwarp_attention_transform_lut) in the P stage");
  RaggedStepArgs a;
  memset(&a, 0, sizeof(a));
  size_t lds = mask_postproc_lds_bytes();
  int kd = 1;
  const attwarp_ragged_header* any = nullptr;
  auto stage = [&](const void* host, const void* dev, const char* which) -> const attwarp_ragged_header* {
    if (!dev) { fail(ATTWARP_E_ARG, "mask_chain_ragged: the %s stage has a host table but no device table", which); return nullptr; }
    const attwarp_ragged_header* h = checked_header(host, which);
    if (!h) return nullptr;
    if (any && (h->C != any->C || h->g != any->g || h->H_out != any->H_out || h->W_out != any->W_out)) {
      fail(ATTWARP_E_ARG, "mask_chain_ragged: the batches of one launch must share C, g and the output size");
      return nullptr;
    }
    any = h;
    lds = std::max(lds, (size_t)h->lds_bytes);
    return h;
  };
  if (r_host) {
    const attwarp_ragged_header* h = stage(r_host, r_dev, "R");
    if (!h) return ATTWARP_E_ARG;
    ATTWARP_REQUIRE(out && map_x && map_y, "mask_chain_ragged: null pointer in the R stage");
    a.r_img = reinterpret_cast<const RaggedImage*>(static_cast<const uint8_t*>(r_dev) + h->off_images);
    a.r_order = reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(r_dev) + h->off_order);
    a.out = out; a.map_x = map_x; a.map_y = map_y;
    a.r_rows = h->rows_per_block; a.r_nblk = h->blocks_per_image;
    a.ord.nR = h->nR;
    kd = h->kd;
  }
  if (f_host) {
    const attwarp_ragged_header* h = stage(f_host, f_dev, "F");
    if (!h) return ATTWARP_E_ARG;
    ATTWARP_REQUIRE(sums_in && map_x_next && map_y_next, "mask_chain_ragged: null pointer in the F stage");
    ATTWARP_REQUIRE(map_x_next != map_x && map_y_next != map_y, "mask_chain_ragged: the next maps must not alias the current ones");
    const uint8_t* t = static_cast<const uint8_t*>(f_dev);
    a.f_img = reinterpret_cast<const RaggedImage*>(t + h->off_images);
    a.f_plans = reinterpret_cast<const RaggedPlan*>(t + h->off_plans);
    a.f_order = reinterpret_cast<const uint32_t*>(t + h->off_order);
    a.sums_in = static_cast<const double*>(sums_in); a.map_x_next = map_x_next; a.map_y_next = map_y_next;
    a.transform = transform; a.exp_scale = exp_scale; a.exp_divisor = exp_divisor; a.apply_inverse = apply_inverse ? 1 : 0;
    a.ord.nF = 2 * h->B;
  }
  if (p_host) {
    const attwarp_ragged_header* h = stage(p_host, p_dev, "P");
    if (!h) return ATTWARP_E_ARG;
    ATTWARP_REQUIRE(mota_in && sums_out, "mask_chain_ragged: null pointer in the P stage");
    ATTWARP_REQUIRE(sums_out != sums_in, "mask_chain_ragged: a stage's output buffer must not alias the buffer the next stage reads in the same launch");
    const uint8_t* t = static_cast<const uint8_t*>(p_dev);
    a.p_img = reinterpret_cast<const RaggedImage*>(t + h->off_images);
    a.p_plans = reinterpret_cast<const RaggedPlan*>(t + h->off_plans);
    a.p_map = reinterpret_cast<const uint32_t*>(t + h->off_pmap);
    a.mota_in = mota_in; a.sums_out = static_cast<double*>(sums_out);
    a.p_lut = transform > ATTWARP_T_SQUARE ? transform_lut : nullptr;
    a.p_square = transform == ATTWARP_T_SQUARE;
    if (a.p_lut) lds = std::max(lds, profiles_u8_lds_bytes<ATTWARP_T_LUT>());
    a.ord.nP = h->nP;
  }
  if (l_host) {
    const attwarp_ragged_header* h = stage(l_host, l_dev, "L");
    if (!h) return ATTWARP_E_ARG;
    ATTWARP_REQUIRE(rev_in && mota_out, "mask_chain_ragged: null pointer in the L stage");
    ATTWARP_REQUIRE(mota_out != mota_in, "mask_chain_ragged: a stage's output buffer must not alias the buffer the next stage reads in the same launch");
    const uint8_t* t = static_cast<const uint8_t*>(l_dev);
    a.l_img = reinterpret_cast<const RaggedImage*>(t + h->off_images);
    a.l_map = reinterpret_cast<const uint32_t*>(t + h->off_lmap);
    a.rev_in = rev_in; a.mota_out = mota_out;
    a.ord.nL = h->nL;
  }
  if (masks) {
    ATTWARP_REQUIRE(rev_out, "mask_chain_ragged: null pointer in the V stage");
    ATTWARP_REQUIRE(rev_out != rev_in, "mask_chain_ragged: a stage's output buffer must not alias the buffer the next stage reads in the same launch");
    ATTWARP_REQUIRE(B_masks > 0 && g > 0, "mask_chain_ragged: non-positive size in the V stage");
    ATTWARP_REQUIRE(kernel_size > 0 && (kernel_size & 1), "mask_chain_ragged: kernel_size must be odd (got %d)", kernel_size);
    if (g > 32 || kernel_size > 7 || B_masks > 65535) return fail(ATTWARP_E_UNSUPPORTED, "mask_chain_ragged: g <= 32, kernel_size <= 7, B <= 65535");
    ATTWARP_REQUIRE(!any || any->g == g, "mask_chain_ragged: g=%d does not match the tables (g=%d)", g, any ? any->g : 0);
    a.masks = masks; a.rev_out = rev_out;
    a.ks = kernel_size; a.coe = enhance_coe;
    a.ord.nV = B_masks;
  }
  a.g = any ? any->g : g;
  if (any) { a.C = any->C; a.Ho = any->H_out; a.Wo = any->W_out; }
  a.prio = tune(TUNE_STEP_PRIO) >= 0 ? tune(TUNE_STEP_PRIO) : 0;
  a.ord.nF8 = (a.ord.nF + 7) / 8;
  a.ord.nV8 = (a.ord.nV + 7) / 8;
  build_interleave(a.ord, tune(TUNE_CHAIN_SEQ) >= 0 ? tune(TUNE_CHAIN_SEQ) : CHAIN_ORDER_DEFAULT);
#ifdef ATTWARP_TUNING
  a.trace = trace_buffer();
#endif
  const long long octs = chain_order_octets(a.ord);
  if (octs * 8 > 2147483647LL) return fail(ATTWARP_E_UNSUPPORTED, "mask_chain_ragged: grid too large");
  if (lds > LDS_DEFAULT_MAX) return fail(ATTWARP_E_UNSUPPORTED, "mask_chain_ragged: %zu bytes of LDS per workgroup (> %zu)", lds, LDS_DEFAULT_MAX);
  if (octs == 0) return ATTWARP_OK;
  hipStream_t st = as_stream(stream);
  const unsigned grid = (unsigned)(octs * 8);
  switch (kd) {
    case 1: return launch_ragged<1>(a, lds, grid, st);
    case 2: return launch_ragged<2>(a, lds, grid, st);
    case 3: return launch_ragged<3>(a, lds, grid, st);
    default: return launch_ragged<4>(a, lds, grid, st);
  }
}
