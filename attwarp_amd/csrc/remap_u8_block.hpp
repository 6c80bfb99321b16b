// The integer cv2 resample of uint8 images (the default of the main_batched chain, AGW/new_method.py:268-271) as a device
// BLOCK: body shared by remap_rows_u8i_kernel (remap_u8.hip) and the one-launch chain step (chain_step.hip).
#pragma once
#include "common.hpp"

namespace attwarp {
namespace u8k {

constexpr int NT = 256;
constexpr int RMAX = 64;

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
  int wpi;   // integer cv2 kernel: workgroups per image; workgroup j owns row blocks j, j + wpi, ... (else == nblk)
  int grp;   // integer cv2 kernel, block order: 0 = contiguous range per XCD, 1 = plain, g >= 2 = XCDs interleaved in groups of g
  int ntiles;              // TILED: column tiles per row (each KO*NT output bytes), else 1
  int map_div;             // maps belong to image b / map_div (planes of a planar image dispatched as images)
  int unaligned;           // integer cv2 kernel: rows / images that are not dword aligned (the UA instantiation)
};

// ---- CV2 mode, integer form (rows of <= 4096 bytes) ----------------------------------------------------------------
// OpenCV's uint8 path is exact integer arithmetic (see the header of this file), so it does not need the float
// pipeline above.  This kernel keeps the same decomposition (workgroup = R output rows of one image, vertical pass
// into LDS, horizontal gather) but
//   * vertical: a thread owns dwords of the two source rows; bytes (0,2) and (1,3) of a dword are two packed 16-bit
//     pairs, v = (32-ky)*top + ky*bottom <= 8160 is one v_pk_mul_lo_u16 + one v_pk_mad_u16 per pair -- 10 VALU
//     instructions per 4 source bytes (the float form: 4 + 4 conversions and 12 lerp operations) -- and the LDS row holds
//     16-bit values (half the LDS bytes), in the order [e0, e2, e1, e3] per group of 4 (no re-interleaving: the tap
//     offsets know the order);
//   * horizontal: a lane produces 4 CONSECUTIVE output bytes: 8 ds_read_u16, per byte
//     ((32 - kx)*v0 + kx*v1 + 512) >> 10 (one v_dot2_u32_u16 on the tap pair, weights scaled by 64 so that the byte is
//     byte 2 of the result), packed with two v_perm_b32 and stored as one dword -- no LDS output row, no flush pass;
//     tap 1 is always "tap 0's pixel + 1" with kx forced to 0 where OpenCV clamps both taps to the same pixel (integer
//     arithmetic: a zero weight is exact), so the clamped cases need no second offset logic.
// Bit-identical to blend<uint8_t, CV2> of remap.hip / the oracle.
// Output dwords are written once and never read by this kernel: nontemporal stores (cache-policy bit 1 of the buffer
// instruction) leave L2 / Infinity Cache to the source rows.  Measured in one process (tools/u8_nt_probe.py, B=256):
// 336 -> 500 95.8 -> 76.2 us, 1024 -> 1024 343.6 -> 336.4, 1024 -> 500 156.0 -> 155.0; nontemporal LOADS lose
// everywhere (the row shared with the neighbouring output row then misses: 1024 -> 1024 459.7 us).  The float32 kernels
// do not respond to either (docs/experiments.md).
constexpr int U8I_STORE_NT = 2;
constexpr int U8I_VLP = 4096 + 16;      // u16 elements per LDS row buffer: rows of <= 4096 bytes + one pixel of slack

constexpr size_t u8i_lds_bytes() { return (size_t)RMAX * sizeof(float) + 2 * (size_t)U8I_VLP * sizeof(uint16_t); }
// One workgroup's share of ONE image: row blocks rb0, rb0 + p.wpi, ... of the image whose pixels start at src_b (img_bytes
// long) and whose output starts at dst_b; mx_b / my_b: that image's maps.  p carries the geometry (H, W, Ho, Wo, CS,
// row_len, VL, ..., R, nblk, wpi); p.src / p.dst / p.mx / p.my / p.img_stride are not read here, so the ragged chain
// (chain_ragged.hip) can fill a Params per image from its descriptor table.  smem: u8i_lds_bytes() of LDS.
// UA ("unaligned", interleaved images only): rows whose byte length is not a multiple of 4 and images that start anywhere
// -- e.g. 683 x 3 bytes per row, the portrait TextVQA case.  gfx950 does serve dword loads at any byte address, but at a
// price: a wave's 64 misaligned dwords cost ~40 % of this kernel's time (tools/attic/ua_cost.py: 684-wide images from a view
// that starts one byte in: 0.192 ms against 0.138 ms).  So the loads stay ALIGNED: the buffer descriptor starts at the
// image's base rounded down to a dword, a row's scalar offset is rounded down likewise, every thread loads the aligned
// dword its row dword starts in AND the next one, and one v_alignbyte_b32 with the row's (block-uniform) byte shift puts
// the row dword together when the row is consumed.  Nothing behind the image's last aligned dword is read (the descriptor
// ends there: a dword that begins behind it returns 0 and only ever supplies bytes behind the row's end).  The last output
// dword of a row is stored byte by byte; output dwords in front of it are dword stores at the row's own offset.
template <int KI, int KD, bool HWC, int PD, bool UA>
__device__ __forceinline__ void remap_rows_u8i_rows(const Params& p, const uint8_t* src_b, int img_bytes, uint8_t* dst_b,
                                                    int oimg_bytes, const float* mx_b, const float* my_b, int rb0, float* smem) {
  static_assert(!UA || HWC, "unaligned rows: interleaved images only (planar ones are dispatched plane by plane)");
  float* s_my = smem;                                               // RMAX
  // two row buffers a COMPILE-TIME distance apart: tap offsets then fold into the 16-bit offset field of the LDS
  // instructions (with a run-time stride every one of the 8 reads per output dword cost a v_add_u32 for its address)
  uint16_t* vrow0 = reinterpret_cast<uint16_t*>(smem + RMAX);
  uint16_t* vrow1 = vrow0 + U8I_VLP;
  const int tid = threadIdx.x;
  // per-image buffer descriptors (block uniform; images are < 2 GiB: plane_stride * NP is checked by the host)
  const unsigned base_sh = UA ? (unsigned)(reinterpret_cast<uintptr_t>(src_b) & 3u) : 0u;      // block uniform
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(src_b) - base_sh, 0, UA ? (int)((img_bytes + base_sh + 3u) & ~3u) : img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdst = __builtin_amdgcn_make_buffer_rsrc(dst_b, 0, oimg_bytes, 0x00020000);

  // source dwords this thread owns (clamped: padding lanes repeat the last dword)
  unsigned goff[KI];                // byte offset inside the image (row 0): unsigned, so the loads take the SGPR-base form
  int voff[KI];                     // u16 index in the LDS row
  {
    const int dpr = p.row_len >> 2, nd = UA ? (p.VL + 3) >> 2 : p.VL >> 2;
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      const int d = min(tid + NT * k, nd - 1);
      const int pl = HWC ? 0 : d / dpr;
      goff[k] = (unsigned)(pl * p.plane_stride) + 4u * (unsigned)(d - pl * dpr);
      voff[k] = 4 * d;
    }
  }
  // output dwords this thread produces; per byte: LDS byte offsets of the two taps (u16 elements in [e0,e2,e1,e3]
  // order) and kx
  unsigned t0[KD][4], t1[KD][4], wpk[KD][4];    // wpk: (32 - kx) | kx << 16, the two weights of v_dot2_u32_u16
  int soff[KD];
  {
    const int dpo = p.orow_len >> 2, ndo = UA ? (p.OVL + 3) >> 2 : p.OVL >> 2;
    // u16 element e of the row lives at position [e0, e2, e1, e3] of its group of four (the vertical pass produces the
    // pairs (0,2) and (1,3) of a source dword; no re-interleaving).  (Swapping the two dwords of a group in every other
    // 64-dword window, so that lanes l and l + 32 of a slope-1 gather use different banks, was measured: no effect --
    // the kernel is bound by VALU issue -- and cost two selects per source dword; taken out.)
    // t0 / t1 hold ABSOLUTE LDS addresses (of the tap in row buffer 0): as offsets from the buffer pointer every one of
    // the 8 reads per output dword paid a v_add_u32 with the (link-time) base of the dynamic LDS block.
    const unsigned vbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)vrow0;
    auto lds_off = [vbase](unsigned e) -> unsigned { return vbase + 2u * ((e & ~3u) | (((e & 1u) << 1) | ((e >> 1) & 1u))); };
#pragma unroll
    for (int k = 0; k < KD; ++k) {
      const int d = min(tid + NT * k, ndo - 1);
      const int pl = HWC ? 0 : d / dpo;
      const int r0 = 4 * (d - pl * dpo);                            // first byte of the dword inside its plane row
      soff[k] = (int)(pl * p.oplane_stride) + r0;
      // (x, c) of the dword's four bytes: ONE division per dword, then a carry per byte (the division of every byte by
      // the run-time channel count was 240 of the ~600 instructions of this prologue, which is half of a workgroup's
      // work at 336 x 336 -> 500 x 500: 16 rows per block)
      int xj = r0 / p.CS, cj = r0 - xj * p.CS;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int x = UA ? min(xj, p.Wo - 1) : xj, c = cj;    // (UA: the bytes behind the row's end in its last dword)
        if (++cj >= p.CS) { cj = 0; ++xj; }
        const float m = mx_b[x];
        const int q = cv_round_q5(m);                               // cvRound
        const int i = q >> 5;
        const int i0 = min(max(i, 0), p.W - 1), i1 = min(max(i + 1, 0), p.W - 1);
        const unsigned kx = (i0 == i1) ? 0u : (unsigned)(q & 31);  // both taps on one pixel: weight of tap 1 is moot
        const unsigned e0 = (unsigned)(pl * p.row_len + i0 * p.CS + c);
        const unsigned e1 = (i0 == i1) ? e0 : (unsigned)(pl * p.row_len + i1 * p.CS + c);
        t0[k][j] = lds_off(e0);
        t1[k][j] = lds_off(e1);
        // opaque, or address-mode sinking moves the "+ base" back in front of every read of the row loop
        asm volatile("" : "+v"(t0[k][j]), "+v"(t1[k][j]));
        wpk[k][j] = ((32u - kx) << 6) | (kx << 22);      // both weights x 64: the result byte lands on a byte boundary
      }
    }
  }

  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) const uint16_t lds_cu16;
  auto row_taps = [&](float m, int& i0, int& i1, unsigned& ky) {
    const int q = cv_round_q5(m);
    const int i = q >> 5;
    i0 = min(max(i, 0), p.H - 1);
    i1 = min(max(i + 1, 0), p.H - 1);
    ky = (unsigned)(q & 31);
  };
  int ci0, ci1;
  // PD register sets: the two source rows of output row q sit in set q % PD, requested PD rows ahead
  unsigned ky[PD];                  // their vertical fractions
  uint32_t A[PD][KI], C[PD][KI];
  uint32_t A1[PD][UA ? KI : 1], C1[PD][UA ? KI : 1];      // UA: the aligned dword behind A / C
  unsigned sha[PD], shc[PD];        // UA: byte shift of the top / bottom row against its aligned start (block uniform)
  int y0 = 0, nrows = 0;
  /* buffer loads / stores: image base in an SGPR descriptor, row offset in the scalar offset, the thread's dword in the
     32-bit vector offset -- no per-access 64-bit address arithmetic (a v_lshl_add_u64 per load and store before) */
#define ATTWARP_U8I_FETCH(AX, CX, AX1, CX1, SA, SC)                                                    \
  {                                                                                                    \
    int ra_ = ci0 * p.row_len, rc_ = ci1 * p.row_len;            /* block uniform: SGPRs */              \
    if (UA) {                                                                                          \
      ra_ += (int)base_sh; rc_ += (int)base_sh;                                                        \
      SA = (unsigned)ra_ & 3u; SC = (unsigned)rc_ & 3u;                                                \
      ra_ &= ~3; rc_ &= ~3;                                                                            \
    }                                                                                                  \
    _Pragma("unroll") for (int k = 0; k < KI; ++k) {                                                    \
      AX[k] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)goff[k], ra_, 0);                         \
      CX[k] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)goff[k], rc_, 0);                         \
      /* (a row that happens to start on a dword boundary -- every fourth row of a 683-pixel image -- needs no second   \
         dword: v_alignbyte_b32 with shift 0 returns the low operand whatever the stale high one holds) */ \
      if (UA && SA) AX1[k] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)goff[k] + 4, ra_, 0);      \
      if (UA && SC) CX1[k] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)goff[k] + 4, rc_, 0);      \
    }                                                                                                  \
  }
  /* One output row.  Its two source rows were requested PD rows ahead (a workgroup's own row takes about a microsecond
     with eight of them sharing a CU, less than the loaded HBM latency: with PD = 1 the kernel waits for memory in
     every row); the vertical pass writes LDS row buffer q & 1, one barrier, the horizontal pass gathers from it.
     Horizontal pass: (32 - kx) * v0 + kx * v1 + 512 is one v_dot2_u32_u16 on the tap pair.  Measured and dropped: both
     taps into one register with ds_read_u16_d16 / _d16_hi -- on this part (SRAM ECC) a d16 load clears the other half
     instead of preserving it. */
#define ATTWARP_U8I_ROW(q_, vbuf, VOFF, AX, CX, KY, AX1, CX1, SA, SC)                                  \
  {                                                                                                    \
    const unsigned w1_ = KY, w0_ = 32u - KY;                                                           \
    if (UA) {   /* the row dwords out of the aligned pairs: (hi:lo) >> 8 * shift */                    \
      _Pragma("unroll") for (int k = 0; k < KI; ++k) {                                                  \
        AX[k] = __builtin_amdgcn_alignbyte(AX1[k], AX[k], SA);                                         \
        CX[k] = __builtin_amdgcn_alignbyte(CX1[k], CX[k], SC);                                         \
      }                                                                                                \
    }                                                                                                  \
    const us2 w0p_ = {(unsigned short)w0_, (unsigned short)w0_}, w1p_ = {(unsigned short)w1_, (unsigned short)w1_}; \
    _Pragma("unroll") for (int k = 0; k < KI; ++k) {                                                    \
      /* bytes (0, 2) and (1, 3) zero-extended to two u16: an and, and ONE v_perm_b32 (shift + and before) */ \
      const uint32_t a02_ = AX[k] & 0x00ff00ffu, a13_ = __builtin_amdgcn_perm(0u, AX[k], 0x0c030c01u);   \
      const uint32_t c02_ = CX[k] & 0x00ff00ffu, c13_ = __builtin_amdgcn_perm(0u, CX[k], 0x0c030c01u);   \
      const us2 v02_ = __builtin_bit_cast(us2, a02_) * w0p_ + __builtin_bit_cast(us2, c02_) * w1p_;      \
      const us2 v13_ = __builtin_bit_cast(us2, a13_) * w0p_ + __builtin_bit_cast(us2, c13_) * w1p_;      \
      uint2 st_;                                                                                       \
      st_.x = __builtin_bit_cast(uint32_t, v02_);                                                      \
      st_.y = __builtin_bit_cast(uint32_t, v13_);                                                      \
      *reinterpret_cast<uint2*>((vbuf) + voff[k]) = st_;                                               \
    }                                                                                                  \
    if ((q_) + PD < nrows) { /* this register set is free: request the rows of output row q + PD */   \
      row_taps(s_my[(q_) + PD], ci0, ci1, KY);                                                         \
      ATTWARP_U8I_FETCH(AX, CX, AX1, CX1, SA, SC)                                                      \
    }                                                                                                  \
    __syncthreads();                                                                                   \
    const int orow_ = (y0 + (q_)) * p.orow_len;                                                        \
    _Pragma("unroll") for (int k = 0; k < KD; ++k) {                                                    \
      unsigned v0_[4], v1_[4];                                                                         \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                   \
        v0_[j] = *(lds_cu16*)(uintptr_t)(t0[k][j] + (VOFF));                                           \
        v1_[j] = *(lds_cu16*)(uintptr_t)(t1[k][j] + (VOFF));                                           \
      }                                                                                                \
      /* 64 * ((32 - kx) * v0 + kx * v1 + 512) < 2^24: one v_dot2_u32_u16 with the weights scaled by 64, so that     \
         ((..) >> 10) -- the output byte -- is BYTE 2 of the result; four results are packed with two v_perm_b32 and  \
         an or (a shift and a shift-or per byte before) */                                             \
      unsigned r_[4];                                                                                  \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                   \
        const us2 pr_ = {(unsigned short)v0_[j], (unsigned short)v1_[j]};                               \
        r_[j] = __builtin_amdgcn_udot2(pr_, __builtin_bit_cast(us2, wpk[k][j]), 512u << 6, false);      \
      }                                                                                                \
      const unsigned o_ = __builtin_amdgcn_perm(r_[1], r_[0], 0x0c0c0602u) | __builtin_amdgcn_perm(r_[3], r_[2], 0x06020c0cu); \
      if (tid + NT * k < (p.OVL >> 2)) __builtin_amdgcn_raw_buffer_store_b32(o_, rdst, soff[k], orow_, U8I_STORE_NT); \
      else if (UA && k == KD - 1 && tid + NT * k == (p.OVL >> 2)) {   /* the row's last 1..3 bytes */           \
        for (int j_ = 0; j_ < (p.OVL & 3); ++j_)                                                       \
          __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(o_ >> (8 * j_)), rdst, soff[k] + j_, orow_, U8I_STORE_NT); \
      }                                                                                                \
    }                                                                                                  \
  }
  // Row blocks of this workgroup: rb0, rb0 + wpi, ... -- the column-tap prologue above (a cvRound and two integer
  // divisions per output byte) is paid once for all of them, while the workgroups of an image sweep it together as
  // one compact window of rows (DRAM page locality), as in remap_rows_kernel.hpp.
  for (int rb = rb0; rb < p.nblk; rb += p.wpi) {
    y0 = rb * p.R;
    nrows = min(y0 + p.R, p.Ho) - y0;
    if (rb != rb0) __syncthreads();          // the previous block's last gather is done with s_my and the row buffers
    if (tid < nrows) s_my[tid] = my_b[y0 + tid];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PD; ++u)
      if (u < nrows) {
        row_taps(s_my[u], ci0, ci1, ky[u]);
        ATTWARP_U8I_FETCH(A[u], C[u], A1[u], C1[u], sha[u], shc[u])
      }
    int q = 0;
    for (; q + 3 < nrows; q += 4) {          // unrolled by 4: LDS buffer q & 1 and register set q % PD are compile-time
      ATTWARP_U8I_ROW(q, vrow0, 0u, A[0], C[0], ky[0], A1[0], C1[0], sha[0], shc[0])
      ATTWARP_U8I_ROW(q + 1, vrow1, 2u * U8I_VLP, A[1 % PD], C[1 % PD], ky[1 % PD], A1[1 % PD], C1[1 % PD], sha[1 % PD], shc[1 % PD])
      ATTWARP_U8I_ROW(q + 2, vrow0, 0u, A[2 % PD], C[2 % PD], ky[2 % PD], A1[2 % PD], C1[2 % PD], sha[2 % PD], shc[2 % PD])
      ATTWARP_U8I_ROW(q + 3, vrow1, 2u * U8I_VLP, A[3 % PD], C[3 % PD], ky[3 % PD], A1[3 % PD], C1[3 % PD], sha[3 % PD], shc[3 % PD])
    }
    if (q < nrows) ATTWARP_U8I_ROW(q, vrow0, 0u, A[0], C[0], ky[0], A1[0], C1[0], sha[0], shc[0])
    if (q + 1 < nrows) ATTWARP_U8I_ROW(q + 1, vrow1, 2u * U8I_VLP, A[1 % PD], C[1 % PD], ky[1 % PD], A1[1 % PD], C1[1 % PD], sha[1 % PD], shc[1 % PD])
    if (q + 2 < nrows) ATTWARP_U8I_ROW(q + 2, vrow0, 0u, A[2 % PD], C[2 % PD], ky[2 % PD], A1[2 % PD], C1[2 % PD], sha[2 % PD], shc[2 % PD])
  }
#undef ATTWARP_U8I_ROW
#undef ATTWARP_U8I_FETCH
}

// A batch of equally shaped images: bid_in = the workgroup's index in launch order (block % 8 names the XCD it runs on)
template <int KI, int KD, bool HWC, int PD, bool UA = false>
__device__ __forceinline__ void remap_rows_u8i_block(const Params& p, int bid_in, float* smem) {
  int bid = bid_in;
  {     // block order, as in remap_rows_kernel.hpp
    const int n = p.nblocks, q = n >> 3, r = n & 7, xcd = bid & 7, idx = bid >> 3;
    if (p.grp == 0) bid = (xcd < r) ? xcd * (q + 1) + idx : r * (q + 1) + (xcd - r) * q + idx;
    else if (p.grp >= 2) {
      const int g = p.grp, per = 8 * g, grp = idx / g, within = idx - grp * g, cand = grp * per + xcd * g + within;
      bid = cand < (n / per) * per ? cand : bid;
    }
  }
  const int b = bid / p.wpi, rb0 = bid - b * p.wpi, bm = b / p.map_div;
  remap_rows_u8i_rows<KI, KD, HWC, PD, UA>(p, p.src + (long long)b * p.img_stride, (int)p.img_stride,
                                           p.dst + (long long)b * p.oimg_stride, (int)p.oimg_stride,
                                           p.mx + (long long)bm * p.Wo, p.my + (long long)bm * p.Ho, rb0, smem);
}

}  // namespace u8k

// Fills the launch geometry of the integer kernel for a [B,H,W,C] (HWC) or [B,C,H,W] uint8 batch; false when the shape
// takes another kernel (rows wider than 4096 bytes, exact mode: remap_u8.hip decides).
bool u8i_params(u8k::Params& p, const uint8_t* src, uint8_t* dst, int layout, int B, int C, int H, int W, int Ho, int Wo,
                const float* mx, const float* my);

}  // namespace attwarp
