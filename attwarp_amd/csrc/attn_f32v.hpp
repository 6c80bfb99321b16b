// A1 on rows with unit kv stride, four tokens per lane (device block shared by attn.hip and the fused step kernel,
// remap_rows_kernel.hpp): BatchMaskHookLogger._process_attention, AGW/attention_extraction/llava.py:385-396.
//
// Accumulation order (restated by oracle/warp_oracle.py::_row_sums_f32_tree / _head_sums_f32 and oracle/warp_ref.c;
// every kernel of A1 produces exactly this): the reference reduces on the model's GPU, where torch accumulates
// float16 and float32 alike in FLOAT32 in an implementation-defined order (llava.py:392-394).  Here:
//   row sum   token t belongs to lane (t mod 256) / 4; a lane adds its tokens four at a time, (x0 + x1) + (x2 + x3),
//             block after block of 256 tokens onto a running float32 sum; the 64 lane sums are combined by the
//             butterfly s[l] += s[l ^ o], o = 32, 16, 8, 4, 2, 1; tokens past the slice count as +0;
//   head mean four partial sums P_w over heads w, w+4, w+8, ... (ascending), then ((0 + P_0) + P_1) + P_2) + P_3.
// Every value crossing a reference op boundary is rounded to the model dtype T as before.
#pragma once
#include <type_traits>
#include "common.hpp"

namespace attwarp {

constexpr int ATTN_NT = 256;

// ---- float32 division by a denominator shared by many numerators -------------------------------------------------
// The compiler expands an IEEE float32 `a / d` into
//     ds = div_scale(d, d, a); ns = div_scale(a, d, a); r0 = rcp(ds); e0 = fma(-ds, r0, 1); r1 = fma(e0, r0, r0);
//     q0 = ns * r1; e1 = fma(-ds, q0, ns); q1 = fma(e1, r1, q0); e2 = fma(-ds, q1, ns); q = div_fmas(e2, r1, q1);
//     result = div_fixup(q, d, a)
// (11 instructions, one of them the quarter-rate v_rcp_f32; 16.6 VALU instructions per attention element made this
// kernel VALU bound).  v_div_scale only rescales when an exponent is extreme -- |a| < 2^-103, d denormal or > 2^126,
// a quotient that is denormal or whose exponents differ by >= 96 -- and v_div_fixup only replaces q for zero /
// infinite / NaN operands.  Inside  d in [2^-60, 4],  a == 0 or 2^-100 <= a <= 2^20  the scale factors are 1, so
// r0, e0, r1 depend on d alone and the SAME sequence costs 5 instructions per numerator, bit for bit the result of
// `a / d` (a == 0 gives +0 through the sequence, as div_fixup does).  Anything outside that box takes `a / d` itself.
struct SharedDiv {
  float d, r1;
  __device__ __forceinline__ explicit SharedDiv(float den) : d(den) {
    const float r0 = __builtin_amdgcn_rcpf(den);
    const float e0 = __builtin_fmaf(-den, r0, 1.0f);
    r1 = __builtin_fmaf(e0, r0, r0);
  }
  __device__ __forceinline__ float operator()(float a) const {
    const float q0 = fmul(a, r1);
    const float e1 = __builtin_fmaf(-d, q0, a);
    const float q1 = __builtin_fmaf(e1, r1, q0);
    const float e2 = __builtin_fmaf(-d, q1, a);
    return __builtin_fmaf(e2, r1, q1);
  }
  // For a quotient that is rounded to float16 / bfloat16 next (numerator and denominator being values of that type: p and
  // p-bit significands, p = 11 / 8) the first refinement is enough: RN_T(q1) == RN_T(RN_32(a / d)) == RN_T(a / d).
  //  * a / d is either exactly a rounding midpoint m of T (a (p+1)-bit number) or at least |a/d| / (M D) > 2^-23 |a/d|
  //    away from every midpoint (a 2^k - m d is a non-zero integer multiple of one unit of the (2p+1)-bit product; in the
  //    subnormal range of float16 the distance is >= 2^-25 / D > 2^-36 against an error of 2^-38);
  //  * q1 = RN_32(q0 + e1 r1) with e1 = a - d q0 exact: |q1 - a/d| <= (2^-24 + 2^-45) |a/d|, so q1 lies on the same side of
  //    every midpoint as a / d, and when a / d IS a midpoint (representable in float32) the value before rounding is
  //    within 2^-45 of it and q1 equals it exactly -- the tie then breaks as it does for RN_32(a / d) itself.
  // Checked exhaustively for float16 (every non-negative finite a, every d in (0, 4], reciprocal off by -1, 0, +1 ulp) and
  // bfloat16 by tests/test_oracle_properties.py::test_three_step_division_* against the float32 quotient.
  __device__ __forceinline__ float first_refinement(float a) const {
    const float q0 = fmul(a, r1);
    const float e1 = __builtin_fmaf(-d, q0, a);
    return __builtin_fmaf(e1, r1, q0);
  }
  template <typename T> __device__ __forceinline__ float rounded_to(float a) const;   // RN_T(a / d) as a float
};
template <> __device__ __forceinline__ float SharedDiv::rounded_to<float>(float a) const { return (*this)(a); }
template <> __device__ __forceinline__ float SharedDiv::rounded_to<__half>(float a) const {
  return __half2float(__float2half_rn(first_refinement(a)));
}
template <> __device__ __forceinline__ float SharedDiv::rounded_to<__hip_bfloat16>(float a) const {
  return __bfloat162float(__float2bfloat16(first_refinement(a)));
}
constexpr uint32_t SDIV_NUM_LO = 0x0D800000u;   // 2^-100
constexpr uint32_t SDIV_NUM_HI = 0x49800000u;   // 2^20

template <typename T> __device__ __forceinline__ T add_tiny(T s);   // s + 1e-12 evaluated in dtype T
template <> __device__ __forceinline__ float add_tiny<float>(float s) { return fadd(s, 1e-12f); }
template <> __device__ __forceinline__ __half add_tiny<__half>(__half s) {
  return __float2half_rn(fadd(__half2float(s), 1e-12f));
}
template <> __device__ __forceinline__ __hip_bfloat16 add_tiny<__hip_bfloat16>(__hip_bfloat16 s) {
  return __float2bfloat16(fadd(__bfloat162float(s), 1e-12f));
}

// ---- four consecutive elements per lane: 16-byte loads for float32 (4-byte aligned: the image-token slice starts at
// an arbitrary token), 8-byte loads for float16 / bfloat16 (2-byte aligned); nontemporal -- the rows are read once.
struct F4v {
  float x, y, z, w;
};
template <typename T> __device__ __forceinline__ F4v load4_nt(const T* p);
template <> __device__ __forceinline__ F4v load4_nt<float>(const float* p) {
  typedef float v4f_a4 __attribute__((ext_vector_type(4), aligned(4)));
  const v4f_a4 q = __builtin_nontemporal_load(reinterpret_cast<const v4f_a4*>(p));
  return F4v{q.x, q.y, q.z, q.w};
}
template <> __device__ __forceinline__ F4v load4_nt<__half>(const __half* p) {
  typedef uint16_t v4h_a2 __attribute__((ext_vector_type(4), aligned(2)));
  const v4h_a2 q = __builtin_nontemporal_load(reinterpret_cast<const v4h_a2*>(p));
  return F4v{__half2float(__builtin_bit_cast(__half, (uint16_t)q.x)), __half2float(__builtin_bit_cast(__half, (uint16_t)q.y)),
             __half2float(__builtin_bit_cast(__half, (uint16_t)q.z)), __half2float(__builtin_bit_cast(__half, (uint16_t)q.w))};
}
template <> __device__ __forceinline__ F4v load4_nt<__hip_bfloat16>(const __hip_bfloat16* p) {
  typedef uint16_t v4h_a2 __attribute__((ext_vector_type(4), aligned(2)));
  const v4h_a2 q = __builtin_nontemporal_load(reinterpret_cast<const v4h_a2*>(p));
  return F4v{__uint_as_float((uint32_t)q.x << 16), __uint_as_float((uint32_t)q.y << 16),
             __uint_as_float((uint32_t)q.z << 16), __uint_as_float((uint32_t)q.w << 16)};
}
// the same four tokens as loaded (the pipelined form of the block keeps the NEXT heads' rows in this form: 2 registers
// per four 16-bit tokens) and their conversion
template <typename T> struct Raw4 { uint32_t lo, hi; };
template <> struct Raw4<float> { float x, y, z, w; };
template <typename T> __device__ __forceinline__ Raw4<T> load4_raw(const T* p) {
  typedef uint16_t v4h_a2 __attribute__((ext_vector_type(4), aligned(2)));
  const v4h_a2 q = __builtin_nontemporal_load(reinterpret_cast<const v4h_a2*>(p));
  return Raw4<T>{(uint32_t)q.x | ((uint32_t)q.y << 16), (uint32_t)q.z | ((uint32_t)q.w << 16)};
}
template <> __device__ __forceinline__ Raw4<float> load4_raw<float>(const float* p) {
  const F4v q = load4_nt<float>(p);
  return Raw4<float>{q.x, q.y, q.z, q.w};
}
__device__ __forceinline__ F4v cvt4(const Raw4<float>& r) { return F4v{r.x, r.y, r.z, r.w}; }
__device__ __forceinline__ F4v cvt4(const Raw4<__half>& r) {
  return F4v{__half2float(__builtin_bit_cast(__half, (uint16_t)(r.lo & 0xffffu))), __half2float(__builtin_bit_cast(__half, (uint16_t)(r.lo >> 16))),
             __half2float(__builtin_bit_cast(__half, (uint16_t)(r.hi & 0xffffu))), __half2float(__builtin_bit_cast(__half, (uint16_t)(r.hi >> 16)))};
}
__device__ __forceinline__ F4v cvt4(const Raw4<__hip_bfloat16>& r) {
  return F4v{__uint_as_float(r.lo << 16), __uint_as_float(r.lo & 0xffff0000u), __uint_as_float(r.hi << 16),
             __uint_as_float(r.hi & 0xffff0000u)};
}
// round a float32 intermediate to the model dtype and back (identity for float32)
template <typename T> __device__ __forceinline__ float round_to(float v) { return to_f32<T>(from_f32<T>(v)); }
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }

template <typename T>
struct AttnStepArgsT {
  const T* attn;           // rows: sample b at attn + b*sb + row_off, head h at + h*sh, unit kv stride
  int heads;
  int64_t sb, sh, row_off;
  const int32_t* starts;   // [starts_mod]; sample b uses starts[b % starts_mod]
  int starts_mod, max_start, ntok;
  T* out;                  // [nb, ntok]
};
typedef AttnStepArgsT<float> AttnStepArgs;
// the same arguments with the dtype as a run-time value (the fused step kernel switches on it, block uniform)
struct AttnStepArgsAny {
  const void* attn;
  int dtype;               // ATTWARP_F32 / F16 / BF16
  int heads;
  int64_t sb, sh, row_off;
  const int32_t* starts;
  int starts_mod, max_start, ntok;
  void* out;
  template <typename T>
  __device__ __forceinline__ AttnStepArgsT<T> as() const {
    AttnStepArgsT<T> a;
    a.attn = static_cast<const T*>(attn); a.heads = heads; a.sb = sb; a.sh = sh; a.row_off = row_off; a.starts = starts;
    a.starts_mod = starts_mod; a.max_start = max_start; a.ntok = ntok; a.out = static_cast<T*>(out);
    return a;
  }
};
// LDS the block needs: (NT / 64) waves x NV*4*64 tokens, float32
template <int NV>
constexpr size_t attn_v4_lds_bytes() { return (size_t)(ATTN_NT / WAVE) * NV * 4 * WAVE * sizeof(float); }

// One 256-thread workgroup per (pseudo-)sample b.  Wave w takes heads w, w+4, ...; lane l owns tokens 4l..4l+3 (+256
// per vector, NV = ceil(ntok / 256)); HU heads are in flight per wave.  part: LDS, attn_v4_lds_bytes<NV>().
// (Measured and dropped: requesting the rows of the next HU heads before the current HU heads are reduced -- two register
// sets of raw rows -- 53.0-53.9 us against 52.0-52.2 for float16 rows.  Round 4: the 64-token tails of the four heads in
// flight sharing ONE vector (lane 16 u + j = tail chunk j of head u; rows moved to lanes 0..15 with v_permlane16/32_swap so
// that the summation order is unchanged; bit-identical): SQ_INSTS_VALU 21.42 M -> 21.22 M per launch for float16 rows at
// T*B = 5120 -- the row moves, per-lane denominators and per-lane row addresses cost what the three saved vectors gain --
// and 55.7-57.6 us against 55.1-57.0 on the same lease: docs/experiments.md.)
template <typename T, int NV, int HU>
__device__ __forceinline__ void attn_reduce_v4_block(const AttnStepArgsT<T>& a, int b, float* part) {
  constexpr int NT = ATTN_NT, PW = NV * 4 * WAVE, NW = NT / WAVE;
  const int heads = a.heads, ntok = a.ntok;
  const int64_t sh = a.sh;
  // the wave index as a scalar: head offsets and row pointers then live in SGPRs (as a vector value every head paid a
  // 64-bit multiply and four 64-bit adds for its three load addresses)
  const int lane = threadIdx.x & (WAVE - 1), wid = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
  // a slice start outside [0, kv_len - ntok] is clamped for memory safety; the host shims reject such starts
  // (the reference would raise: a truncated slice cannot be stacked, llava.py:390-395)
  const int st = min(max(a.starts[b % a.starts_mod], 0), a.max_start);
  const T* base = a.attn + (int64_t)b * a.sb + a.row_off + st;
  float acc[NV][4];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.0f;

  // The image-token slice starts at an arbitrary token (35 + ... in LLaVA-1.5 prompts), so the natural load of a lane --
  // its four tokens -- is only element aligned.  For the 16-bit dtypes an ODD token offset makes it a 2-byte aligned 8-byte
  // load, which the texture addresser splits: float16 rows 59.6 us against 47.0 us for an aligned slice (T*B = 5120 rows x
  // 32 heads, tools/attic/attn_layout_probe.py; a dword-aligned address is as good as an 8-byte aligned one, and float32 rows
  // do not care).  In that case a lane loads the four-token chunk one token BELOW its first token (dword aligned: chunk
  // c = 64 i + l of the row counted from there) and assembles its tokens from its own chunk and the first token pair of
  // the next lane's in registers -- same values in the same lanes, only the way they get there: 59.6 -> 48.0 us.  The
  // first chunk then begins one token before the slice (an odd address is never the first element of an allocation) and
  // the last one ends up to three tokens behind it: tokens of the same row (`tail_room` below).
  auto head_ptr = [&](int h) { return base + (int64_t)min(h, heads - 1) * sh; };
  // tokens by which a row's slice misses the alignment its loads need: a dword-aligned 8-byte load is as good as an
  // 8-byte aligned one, so only an ODD token offset (a 2-byte aligned address) is realigned
  // (only while three tokens behind the slice still belong to the row -- the last chunk reads up to there; a slice that
  // ends within three tokens of the row's end keeps the plain element-aligned loads, which read exactly the slice)
  // ... and only while lane 63 of the LAST vector owns no token: at ntok == NV * 256 its fourth token would come from
  // chunk NV * 64, which no vector loads (wave_rol would hand it chunk (NV - 1) * 64's first dword instead)
  const bool tail_room = st + 3 <= a.max_start && ntok < NV * 4 * WAVE;
  auto misalign = [&](const T* rp) { return tail_room ? (int)((reinterpret_cast<uintptr_t>(rp) / sizeof(T)) & 1u) : 0; };
  constexpr bool ALIGN16 = sizeof(T) == 2;
  auto load = [&](Raw4<T> (&r)[HU][NV], int h0) {
#pragma unroll
    for (int u = 0; u < HU; ++u) {
      const T* rp = head_ptr(h0 + u * NW);
      if constexpr (ALIGN16) {
        const int mis = misalign(rp);
        const T* ap = rp - mis;                                     // dword aligned
        const int cmax = (mis + ntok - 1) >> 2;                     // last four-token chunk that holds a token of the slice
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          unsigned bo = (unsigned)min(lane + WAVE * i, cmax) * 4u * (unsigned)sizeof(T);
          asm volatile("" : "+v"(bo));    // the zero extension stays next to the load: scalar base + 32-bit lane offset
          // dword aligned when realigned (mis == 1) or when the slice starts on an even token; a slice that starts on an odd
          // token without tail room keeps its 2-byte aligned address: the type must not promise more than that
          typedef uint32_t v2u __attribute__((ext_vector_type(2), aligned(2)));
          const v2u q = __builtin_nontemporal_load(reinterpret_cast<const v2u*>(reinterpret_cast<const char*>(ap) + bo));
          r[u][i].lo = q.x;
          r[u][i].hi = q.y;
        }
      } else {                            // float32: the element-aligned 16-byte load itself (realigning gains nothing there)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const int t = min(4 * lane + 4 * WAVE * i, ntok - 4);     // clamped: tail lanes re-read, masked below
          unsigned bo = (unsigned)t * (unsigned)sizeof(T);
          asm volatile("" : "+v"(bo));
          r[u][i] = load4_raw<T>(reinterpret_cast<const T*>(reinterpret_cast<const char*>(rp) + bo));
        }
      }
    }
  };
  // tokens 1 .. 4 of the 8 tokens [own chunk | next chunk].  mis is wave uniform (a scalar branch per head); the next
  // lane's dword comes through DPP wave_shl:1, whose lane 63 keeps `old`: the first chunk of the next vector rotated into
  // lane 63 (wave_rol:1) -- no scalar round trip.
  auto next_dword = [&](uint32_t own, uint32_t next_vec) -> uint32_t {
    const uint32_t rot = __builtin_amdgcn_update_dpp(0u, next_vec, 0x134, 0xf, 0xf, false);     // wave_rol:1: lane 63 <- lane 0
    return __builtin_amdgcn_update_dpp(rot, own, 0x130, 0xf, 0xf, false);                       // wave_shl:1: lane l <- lane l + 1
  };
  auto realign = [&](const Raw4<T> (&r)[NV], int mis, Raw4<T> (&o)[NV]) {
    if (!ALIGN16 || mis == 0) {
#pragma unroll
      for (int i = 0; i < NV; ++i) o[i] = r[i];
      return;
    }
    if constexpr (ALIGN16) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {                                // dwords [w0 w1 | w2 ..], two tokens each; tokens 1 .. 4
        const Raw4<T>& n = r[i + 1 < NV ? i + 1 : i];               // (the last vector's lane 63 holds no token)
        const uint32_t w2 = next_dword(r[i].lo, n.lo);
        o[i].lo = __builtin_amdgcn_alignbyte(r[i].hi, r[i].lo, 2);
        o[i].hi = __builtin_amdgcn_alignbyte(w2, r[i].hi, 2);
      }
    }
  };
  auto reduce = [&](const Raw4<T> (&r)[HU][NV], int h0) {
#pragma unroll
    for (int u = 0; u < HU; ++u) {
      if (h0 + u * NW < heads) {                                    // wave uniform
        F4v v[NV];
        Raw4<T> sh4[NV];
        realign(r[u], misalign(head_ptr(h0 + u * NW)), sh4);
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = cvt4(sh4[i]);
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
          if (4 * lane + 4 * WAVE * i < ntok)
            s = fadd(s, fadd(fadd(v[i].x, v[i].y), fadd(v[i].z, v[i].w)));
        s = wave_sum_dpp(s);                                        // butterfly, o = 32 .. 1
        const float den = to_f32<T>(add_tiny<T>(from_f32<T>(s)));   // (row sum -> T) + 1e-12 in T
        // smallest non-zero and largest bit pattern of this lane's numerators (0 - 1 wraps to the top: zeros do not
        // lower the minimum; negative, infinite and NaN numerators exceed SDIV_NUM_HI)
        // (float16 numerators are 0 or >= 2^-24: only the upper bound -- negative, Inf, NaN -- can fail)
        constexpr bool kNeedLo = !std::is_same<T, __half>::value;
        uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const uint32_t bx = __float_as_uint(v[i].x), by = __float_as_uint(v[i].y),
                         bz = __float_as_uint(v[i].z), bw = __float_as_uint(v[i].w);
          if (kNeedLo) lo = min(min(lo, bx - 1u), min(by - 1u, min(bz - 1u, bw - 1u)));
          hi = max(max(hi, bx), max(by, max(bz, bw)));
        }
        const bool box = den >= 8.673617379884035e-19f && den <= 4.0f &&       // 2^-60 .. 4 (wave uniform)
                         __all(lo >= SDIV_NUM_LO - 1u && hi <= SDIV_NUM_HI);
        if (box) {
          const SharedDiv dv(den);
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            acc[i][0] = fadd(acc[i][0], dv.template rounded_to<T>(v[i].x));
            acc[i][1] = fadd(acc[i][1], dv.template rounded_to<T>(v[i].y));
            acc[i][2] = fadd(acc[i][2], dv.template rounded_to<T>(v[i].z));
            acc[i][3] = fadd(acc[i][3], dv.template rounded_to<T>(v[i].w));
          }
        } else {
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            acc[i][0] = fadd(acc[i][0], round_to<T>(v[i].x / den));
            acc[i][1] = fadd(acc[i][1], round_to<T>(v[i].y / den));
            acc[i][2] = fadd(acc[i][2], round_to<T>(v[i].z / den));
            acc[i][3] = fadd(acc[i][3], round_to<T>(v[i].w / den));
          }
        }
      }
    }
  };
  constexpr int HS = NW * HU;                                       // heads per iteration of a wave
  for (int h0 = wid; h0 < heads; h0 += HS) {                       // (h0 is a scalar: wave-uniform branches)
    Raw4<T> r[HU][NV];
    load(r, h0);
    reduce(r, h0);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) part[wid * PW + 4 * lane + 4 * WAVE * i + j] = acc[i][j];
  __syncthreads();
  const float nheads = to_f32<T>(from_f32<T>((float)heads));
  for (int t = threadIdx.x; t < ntok; t += NT) {
    float m = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; ++w) m = fadd(m, part[w * PW + t]);
    a.out[(int64_t)b * ntok + t] = from_f32<T>(round_to<T>(m) / nheads);    // mean = sum / N in dtype T
  }
}


}  // namespace attwarp
