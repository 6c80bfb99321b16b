// Host side of the float32 staged resample (kernel: remap_rows_kernel.hpp) + the EXACT-mode instantiations.
#include "remap_rows_kernel.hpp"

#include <cstring>

namespace attwarp {

int launch_rows_exact(const RowsParams& p, int tile_ko, hipStream_t st, const StepExtra* ex) {
  if (ex) return launch_step_exact(p, tile_ko, st, ex);
  return launch_rows_mode<ATTWARP_EXACT, false, 1, 4, false>(p, tile_ko, st, nullptr);
}

// Returns via *handled whether the fast path took the request.
// ex != nullptr: the fused step (block ranges for the map construction and the attention reduce of other batches are
// appended to the resample's grid; ex->nR8 is filled in here).
int launch_remap_rows(const float* src, float* dst, int layout, int B, int C, int H, int W, int Ho, int Wo,
                      const float* mx, const float* my, int mode, hipStream_t st, bool* handled, StepExtra* ex) {
  *handled = false;
  if (tune(TUNE_REMAP_VARIANT) == 1) return ATTWARP_OK;  // force the generic gather kernel (A/B measurements, tests)
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
  if (const int se = tune(TUNE_REMAP_CHW_SPLIT); se >= 0) split = layout == ATTWARP_CHW && C > 1 && se != 0;
  // "unaligned": a row length that is not a multiple of 4 floats, or an image / plane that does not start on a 16-byte
  // boundary.  Served by the UA instantiations (remap_rows_ua.hip: interleaved rows that fit the LDS row; planar images
  // plane by plane); the fused step, column-tiled rows and rows of fewer than 4 floats take the generic gather kernel.
  const bool ua = ((long long)W * (layout == ATTWARP_HWC ? C : 1)) % 4 != 0 || (reinterpret_cast<uintptr_t>(src) & 15u) != 0 ||
                  (layout == ATTWARP_HWC ? ((long long)H * W * C) % 4 != 0 : ((long long)H * W) % 4 != 0);
  if (ua && (ex || (reinterpret_cast<uintptr_t>(src) & 3u) != 0)) return ATTWARP_OK;
  if (ua && layout == ATTWARP_CHW && C > 1) {
    if ((long long)B * C > 2147483647LL) return ATTWARP_OK;
    split = true;
  }
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
  // LDS indices are 16 bit and the tap tables live in VGPRs: up to 4096 floats per staged row.  Wider interleaved /
  // one-plane rows are processed in column tiles; wider multi-plane rows take the generic kernel.
  // (tuning flavour, remap_tiled = 2: column tiles also for rows that fit the LDS row -- the experiment of round 5)
  const bool tiled = VL > 4096 || OVL > 4096 || (tune(TUNE_REMAP_TILED) == 2 && p.NP == 1 && !ua && !ex);
  if (ua && (tiled || VL < 4 || p.NP != 1)) return ATTWARP_OK;
  p.unaligned = ua ? 1 : 0;
  if (tiled && (p.NP != 1 || VL > 2147483647LL / 8 || OVL > 2147483647LL / 8)) return ATTWARP_OK;
  if (tiled && tune(TUNE_REMAP_TILED) == 0) return ATTWARP_OK;
  p.VLV = (int)((VL + 3) / 4);
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
  if (const int v = tune(TUNE_REMAP_TILE_KO); v == 8 || v == 12 || v == 4) TILE_KO = v;
  if (tiled) p.ntiles = (int)((OVL + TILE_KO * NT_BIG - 1) / (TILE_KO * NT_BIG));
  const long long row_bytes = tiled ? (long long)TILE_KO * NT_BIG * 4 : VL * 4;
  // (4 KB rows, 336x336x3: R=6 4.85 TB/s, R=12 4.62 at B=256; flat at B=64)
  int R = (int)((24 * 1024 + row_bytes / 2) / row_bytes);
  R = R < 4 ? 4 : (R > 16 ? 16 : R);
  if (tiled) R = 8;
  // Block order and row blocks per workgroup.  Two box / allocation states exist on MI355X: with a
  // contiguous range of row blocks per XCD (group = 0) the eight XCDs stream eight regions ~B/8 images apart, which is
  // the fastest order on some leases (1.09 ms at 1024x1024x3 B=256) and the slowest on others (1.175 ms); with the XCDs
  // interleaved in groups of g row blocks (g - 1 of g halo seams still meet in one L2) all of them work in ONE compact
  // window and the time is the same on both kinds of lease.  Rows of >= 10 KB therefore run R = 3 in groups (shipped:
  // cv2 groups of 8 with two row blocks per workgroup, exact groups of 3 with one -- the history of that choice is below
  // and in docs/experiments.md); the choice is made for the worst case (profiles/round3_order_sweeps.md).  cpw = row
  // blocks per workgroup, strided by the workgroups of the image: the column-tap prologue is paid once per cpw blocks
  // (peaked maps 0.97 -> 0.93 ms, 768x768x3 0.84 -> 0.76).  (LDS: cv2 rows of 8-12 KB use two [top | bottom] buffers
  // and one barrier per row, the fused step and wider rows one buffer and two: remap_rows_kernel.hpp, SINGLE.)
  int group = 0, cpw = 1;
  if (!tiled && row_bytes >= 10 * 1024) {
    // round 4 (profiles/round4_remap_pmc.txt, round4_order_sweep_*.txt): cv2 groups of 8 (first with 3 row blocks per
    // workgroup, shipped with 2: see below) instead of groups of 4 -- the halo rows that cross XCDs drop from 1/4 to 1/8 of the seams (fabric reads
    // 1.090 -> 1.051 x algorithmic, total traffic 1.045 -> 1.025 x), +0.7-1 % on both kinds of lease, peaked maps +3 %;
    // larger groups (12, 16, 24) and more rows per block (5, 6) lose; exact mode is fastest as it was (groups of 3; 8 loses 7 %)
    // (round 4, with the row loop whose look-ahead overlaps and one row list per workgroup: which of 1 / 2 / 3 row blocks per
    //  workgroup is fastest flips with the lease state -- uniform maps, two leases: 1.08 / 1.16 / 1.19 ms and 1.17 / 1.11 / 1.12;
    //  two has the best worst case, docs/experiments.md)
    R = 3;
    group = mode == ATTWARP_CV2 ? 8 : 3;
    cpw = mode == ATTWARP_CV2 ? 2 : 1;
  } else if (!tiled && row_bytes >= 5 * 1024 && mode == ATTWARP_CV2) {
    cpw = 4;
  } else if (!tiled && split) {
    // planes of a planar image as one-channel images (4 KB rows at 1024).  Round 3: groups of 2 (1.170 -> 1.150 ms on a slow
    // lease with the old row loop).  Round 4, the loop whose look-ahead overlaps: the contiguous order on five leases
    // 1.075-1.161 ms against 1.18-1.21 for groups of 2 (R4 g8: 1.12-1.16): docs/experiments.md
    group = 0;
  } else if (!tiled && row_bytes < 5 * 1024 && (long long)B * ((Ho + R - 1) / R) >= 8192) {
    group = 2;        // large batches of small images (336x336x3, B=256): 0.1434 -> 0.1414 ms alone, 0.215 -> 0.209 ms
  }                   // inside the fused step (tools/attic/ab_fused336.py); B=64 is fastest with the contiguous order
  if (const int v = tune(TUNE_REMAP_ROWS); v >= 1 && v <= RMAX) R = v;
  if (R > Ho) R = Ho;
  p.R = R;
  p.nblk = (Ho + R - 1) / R;
  if (const int v = tune(TUNE_REMAP_CPW); v >= 1) cpw = v;
  while (cpw > 1 && (long long)((p.nblk + cpw - 1) / cpw) * B * p.ntiles < 4096) cpw >>= 1;   // keep the chip filled
  if (cpw * R > RMAX) cpw = RMAX / R;        // a workgroup's rows are one list in LDS (remap_rows_block)
  p.wpi = (p.nblk + cpw - 1) / cpw;
  const long long nb = (long long)p.wpi * B * p.ntiles;
  if (nb > 2147483647LL) return ATTWARP_OK;
  p.nblocks = (int)nb;
  p.no_swz = tune(TUNE_REMAP_NOSWZ) >= 0 ? tune(TUNE_REMAP_NOSWZ) : group;
#ifdef ATTWARP_TUNING
  p.alt_dir = tune(TUNE_REMAP_ALT) != 0;
  p.skew = tune(TUNE_REMAP_SKEW) >= 0 ? tune(TUNE_REMAP_SKEW) : 0;
  p.lds_pad = 0;
  if (const int v = tune(TUNE_REMAP_LDSPAD); v >= 0 && v <= 90000) p.lds_pad = v;
  p.trace = trace_buffer();
  p.bound = tune(TUNE_BOUND) > 0 ? (tune(TUNE_BOUND) & 1) : 0;
  p.nt_loads = tune(TUNE_REMAP_NT) > 0 ? tune(TUNE_REMAP_NT) : 0;      // bit 0: nontemporal loads of block-private rows, bit 1: nontemporal stores
#endif
  *handled = true;
  // (one-wave workgroups for rows <= 4 KB were measured too: 336x336x3, B=256: 0.136 ms vs 0.134 ms with
  //  4-wave workgroups -- no gain, so a single workgroup size is instantiated)
  const int tile_ko = tiled ? TILE_KO : 0;
  if (ex) ex->nR8 = (p.nblocks + 7) / 8;
  if (ua) return launch_rows_ua(p, mode, st);
  if (mode == ATTWARP_CV2) return launch_rows_cv2(p, tile_ko, st, ex);
  return launch_rows_exact(p, tile_ko, st, ex);
}

}  // namespace attwarp

using namespace attwarp;

namespace {
// one slot of a fused step: the buffers of the three pieces (a piece is absent when its first pointer is null)
struct SlotPtrs {
  const float* src; float* dst; const float* map_x; const float* map_y;           // R
  const void* steps_in; float* map_x_next; float* map_y_next;                      // M
  const void* rows; const int32_t* starts; void* steps_out;                        // A
};

int step_fused_impl(const SlotPtrs* sl, int nslots, int layout, int B, int C, int H, int W, int H_out, int W_out, int mode,
                    int attn_dtype, int T, int g, const double* inv_x, const double* inv_y, int n_rows, int heads, int kv_len,
                    int starts_mod, int ntok, void* stream) {
  ATTWARP_REQUIRE(nslots == 1 || nslots == 2, "warp_step_fused: nslots must be 1 or 2 (got %d)", nslots);
  const SlotPtrs& s0 = sl[0];
  for (int k = 0; k < nslots; ++k) {
    ATTWARP_REQUIRE(sl[k].src && sl[k].dst && sl[k].map_x && sl[k].map_y, "warp_step_fused: null image / map pointer");
    ATTWARP_REQUIRE((sl[k].steps_in != nullptr) == (s0.steps_in != nullptr) && (sl[k].rows != nullptr) == (s0.rows != nullptr),
                    "warp_step_fused: both slots must carry the same pieces");
  }
  ATTWARP_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && H_out > 0 && W_out > 0, "warp_step_fused: non-positive size");
  ATTWARP_REQUIRE(layout == ATTWARP_HWC || layout == ATTWARP_CHW, "warp_step_fused: unknown layout %d", layout);
  ATTWARP_REQUIRE(mode == ATTWARP_EXACT || mode == ATTWARP_CV2, "warp_step_fused: unknown mode %d", mode);
  if (C > 4 || B > 65535 || H_out > 65535 || (long long)W_out * C > 2147483647LL / 2 ||
      (long long)H * W * C > 2147483647LL)
    return fail(ATTWARP_E_UNSUPPORTED, "warp_step_fused: image shape outside the staged resample's limits");
  if (s0.steps_in || s0.rows)
    ATTWARP_REQUIRE(attn_dtype == ATTWARP_F32 || attn_dtype == ATTWARP_F16 || attn_dtype == ATTWARP_BF16,
                    "warp_step_fused: attn_dtype must be F32, F16 or BF16 (got %d)", attn_dtype);
  if (nslots == 2) {
    // 16-byte vector loads of the resample: the geometry is planned on slot 0, slot 1 must satisfy the same alignment
    ATTWARP_REQUIRE(((reinterpret_cast<uintptr_t>(sl[1].src) ^ reinterpret_cast<uintptr_t>(s0.src)) & 15u) == 0,
                    "warp_step_fused: both slots' images must have the same 16-byte alignment");
    // (an absent piece has null pointers in BOTH slots: only the pieces that are present are compared)
    ATTWARP_REQUIRE(sl[1].dst != s0.dst, "warp_step_fused: the two slots must write different buffers");
    if (s0.steps_in)
      ATTWARP_REQUIRE(sl[1].map_x_next != s0.map_x_next && sl[1].map_y_next != s0.map_y_next,
                      "warp_step_fused: the two slots must write different map buffers");
    if (s0.rows)
      ATTWARP_REQUIRE(sl[1].steps_out != s0.steps_out, "warp_step_fused: the two slots must write different step-map buffers");
  }
  StepExtra ex;
  memset(&ex, 0, sizeof(ex));
  ex.nslots = nslots;
  ex.prio = tune(TUNE_STEP_PRIO) >= 0 ? tune(TUNE_STEP_PRIO) : 0;
#ifdef ATTWARP_TUNING
  ex.trace = trace_buffer();
#endif
  if (s0.steps_in) {      // M: per-step maps of a later batch -> its inverse maps
    ATTWARP_REQUIRE(inv_x && inv_y, "warp_step_fused: null map-construction pointer");
    for (int k = 0; k < nslots; ++k) {
      ATTWARP_REQUIRE(sl[k].map_x_next && sl[k].map_y_next, "warp_step_fused: null map-construction pointer");
      for (int q = 0; q < nslots; ++q)
        ATTWARP_REQUIRE(sl[k].map_x_next != sl[q].map_x && sl[k].map_y_next != sl[q].map_y,
                        "warp_step_fused: the next maps must not alias the current ones");
    }
    ATTWARP_REQUIRE(T > 0 && g > 0, "warp_step_fused: non-positive T / g");
    if (g > 32 || std::max(W, H) > 8192) return fail(ATTWARP_E_UNSUPPORTED, "warp_step_fused: g > 32 or max(W,H) > 8192");
    ex.maps.steps = s0.steps_in; ex.maps.step_dtype = attn_dtype; ex.maps.T = T; ex.maps.B = B; ex.maps.g = g; ex.maps.W = W; ex.maps.H = H;
    ex.maps.W_out = W_out; ex.maps.H_out = H_out; ex.maps.inv_x = inv_x; ex.maps.inv_y = inv_y;
    ex.maps.map_x = s0.map_x_next; ex.maps.map_y = s0.map_y_next; ex.maps.att_out = nullptr;
    ex.nM8 = (nslots * 2 * B + 7) / 8;
  }
  if (s0.rows) {          // A: attention rows of a later batch -> its per-step maps
    for (int k = 0; k < nslots; ++k) {
      ATTWARP_REQUIRE(sl[k].starts && sl[k].steps_out, "warp_step_fused: null attention pointer");
      for (int q = 0; q < nslots; ++q)
        ATTWARP_REQUIRE(sl[k].steps_out != sl[q].steps_in, "warp_step_fused: steps_out must not alias steps_in");
    }
    ATTWARP_REQUIRE(n_rows > 0 && heads > 0 && kv_len > 0 && ntok > 0 && starts_mod > 0, "warp_step_fused: non-positive attention size");
    ATTWARP_REQUIRE(ntok <= kv_len, "warp_step_fused: ntok=%d > kv_len=%d", ntok, kv_len);
    if (ntok % 4 != 0 || ntok > 3 * 4 * WAVE)
      return fail(ATTWARP_E_UNSUPPORTED, "warp_step_fused: ntok must be a multiple of 4 and <= 768");
    // A writes steps_out as [n_rows, ntok]; a later step's M reads that buffer as [T, B, g * g]
    if (s0.steps_in && ntok != g * g)
      return fail(ATTWARP_E_UNSUPPORTED, "warp_step_fused: ntok=%d must equal g*g=%d when both pieces are present", ntok, g * g);
    ex.attn.attn = s0.rows; ex.attn.dtype = attn_dtype; ex.attn.heads = heads; ex.attn.sb = (int64_t)heads * kv_len; ex.attn.sh = kv_len;
    ex.attn.row_off = 0; ex.attn.starts = s0.starts; ex.attn.starts_mod = starts_mod; ex.attn.max_start = kv_len - ntok;
    ex.attn.ntok = ntok; ex.attn.out = s0.steps_out;
    ex.nA = n_rows;
    ex.nA8 = (nslots * n_rows + 7) / 8;
  }
  if (nslots == 2) {
    const SlotPtrs& t = sl[1];
    ex.s1 = StepSlot2{t.src, t.dst, t.map_x, t.map_y, t.steps_in, t.map_x_next, t.map_y_next, t.rows, t.starts, t.steps_out};
  }
  bool handled = false;
  const int rc = launch_remap_rows(s0.src, s0.dst, layout, B, C, H, W, H_out, W_out, s0.map_x, s0.map_y, mode, as_stream(stream),
                                   &handled, &ex);
  if (!handled && rc == ATTWARP_OK)
    return fail(ATTWARP_E_UNSUPPORTED, "warp_step_fused: this image shape / alignment takes the generic resample; use the three separate launches");
  return rc;
}
}  // namespace

extern "C" int attwarp_warp_step_fused(const float* src, float* dst, int layout, int B, int C, int H, int W, int H_out,
                                       int W_out, const float* map_x, const float* map_y, int mode,
                                       int attn_dtype, const void* steps_in, int T, int g, const double* inv_x,
                                       const double* inv_y, float* map_x_next, float* map_y_next,
                                       const void* rows, int n_rows, int heads, int kv_len, const int32_t* starts,
                                       int starts_mod, int ntok, void* steps_out, void* stream) {
  const SlotPtrs s{src, dst, map_x, map_y, steps_in, map_x_next, map_y_next, rows, starts, steps_out};
  return step_fused_impl(&s, 1, layout, B, C, H, W, H_out, W_out, mode, attn_dtype, T, g, inv_x, inv_y, n_rows, heads, kv_len,
                         starts_mod, ntok, stream);
}

extern "C" int attwarp_warp_step_fused_slots(const attwarp_step_slot* slots, int nslots, int layout, int B, int C, int H, int W,
                                             int H_out, int W_out, int mode, int attn_dtype, int T, int g, const double* inv_x,
                                             const double* inv_y, int n_rows, int heads, int kv_len, int starts_mod, int ntok,
                                             void* stream) {
  ATTWARP_REQUIRE(slots, "warp_step_fused_slots: null slot table");
  ATTWARP_REQUIRE(nslots == 1 || nslots == 2, "warp_step_fused_slots: nslots must be 1 or 2 (got %d)", nslots);
  SlotPtrs s[2];
  for (int k = 0; k < nslots; ++k)
    s[k] = SlotPtrs{slots[k].src, slots[k].dst, slots[k].map_x, slots[k].map_y, slots[k].steps_in, slots[k].map_x_next,
                    slots[k].map_y_next, slots[k].rows, slots[k].starts, slots[k].steps_out};
  return step_fused_impl(s, nslots, layout, B, C, H, W, H_out, W_out, mode, attn_dtype, T, g, inv_x, inv_y, n_rows, heads, kv_len,
                         starts_mod, ntok, stream);
}
