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
};
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
  for (int h0 = wid; h0 < heads; h0 += NW * HU) {
    F4v v[HU][NV];
#pragma unroll
    for (int u = 0; u < HU; ++u) {
      const T* rp = base + (int64_t)min(h0 + u * NW, heads - 1) * sh;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int t = min(4 * lane + 4 * WAVE * i, ntok - 4);       // clamped: tail lanes re-read, masked below
        unsigned bo = (unsigned)t * (unsigned)sizeof(T);
        asm volatile("" : "+v"(bo));      // the zero extension stays next to the load: scalar base + 32-bit lane offset
        v[u][i] = load4_nt<T>(reinterpret_cast<const T*>(reinterpret_cast<const char*>(rp) + bo));
      }
    }
#pragma unroll
    for (int u = 0; u < HU; ++u) {
      if (h0 + u * NW < heads) {                                    // wave uniform
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
          if (4 * lane + 4 * WAVE * i < ntok)
            s = fadd(s, fadd(fadd(v[u][i].x, v[u][i].y), fadd(v[u][i].z, v[u][i].w)));
        s = wave_sum_dpp(s);                                        // butterfly, o = 32 .. 1
        const float den = to_f32<T>(add_tiny<T>(from_f32<T>(s)));   // (row sum -> T) + 1e-12 in T
        // smallest non-zero and largest bit pattern of this lane's numerators (0 - 1 wraps to the top: zeros do not
        // lower the minimum; negative, infinite and NaN numerators exceed SDIV_NUM_HI)
        // (float16 numerators are 0 or >= 2^-24: only the upper bound -- negative, Inf, NaN -- can fail)
        constexpr bool kNeedLo = !std::is_same<T, __half>::value;
        uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const uint32_t bx = __float_as_uint(v[u][i].x), by = __float_as_uint(v[u][i].y),
                         bz = __float_as_uint(v[u][i].z), bw = __float_as_uint(v[u][i].w);
          if (kNeedLo) lo = min(min(lo, bx - 1u), min(by - 1u, min(bz - 1u, bw - 1u)));
          hi = max(max(hi, bx), max(by, max(bz, bw)));
        }
        const bool box = den >= 8.673617379884035e-19f && den <= 4.0f &&       // 2^-60 .. 4 (wave uniform)
                         __all(lo >= SDIV_NUM_LO - 1u && hi <= SDIV_NUM_HI);
        if (box) {
          const SharedDiv dv(den);
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            acc[i][0] = fadd(acc[i][0], round_to<T>(dv(v[u][i].x)));
            acc[i][1] = fadd(acc[i][1], round_to<T>(dv(v[u][i].y)));
            acc[i][2] = fadd(acc[i][2], round_to<T>(dv(v[u][i].z)));
            acc[i][3] = fadd(acc[i][3], round_to<T>(dv(v[u][i].w)));
          }
        } else {
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            acc[i][0] = fadd(acc[i][0], round_to<T>(v[u][i].x / den));
            acc[i][1] = fadd(acc[i][1], round_to<T>(v[u][i].y / den));
            acc[i][2] = fadd(acc[i][2], round_to<T>(v[u][i].z / den));
            acc[i][3] = fadd(acc[i][3], round_to<T>(v[u][i].w / den));
          }
        }
      }
    }
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
