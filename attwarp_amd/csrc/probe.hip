// hook-side capture.  The reference obtains the layer-20 attention probabilities by forcing
// output_attentions=True on that layer (AGW/attention_extraction/llava.py:422-438), which makes the eager
// attention materialise softmax(QK^T) for EVERY query row, [B,heads,q,kv], although the hook only consumes the
// LAST row (llava.py:391).  This kernel computes that one row straight from the post-RoPE query of the last
// token and the key cache, and chains into the A1 reduction (attn_reduce_step_kernel): K is read once,
// nothing of size q*kv is ever written.
//
// Arithmetic follows HF's eager_attention_forward (transformers, pinned 4.37.2 in the reference's
// attwarp.yaml:26; not vendored) at its dtype transitions, T = model dtype:
//   w  = matmul(q, k^T)            -> T      exact products, accumulated in double, rounded once to T
//   w  = w * scaling               -> T      float32 multiply, rounded to T
//   p  = softmax(w, dtype=float32) -> T      max, exp(w - max), sum in double, float32 divide, rounded to T
// Left padding: kv positions below kv_begin[b] carry the mask's -inf / finfo.min, i.e. probability 0.
// Then per head  p[st:st+ntok] / (sum + 1e-12)  and the mean over heads in dtype T (llava.py:393), by the
// same kernel the hook path uses, so a probed step equals a hooked step given the same probabilities.
//
// Bandwidth: one workgroup per (head, sample) streams kv rows of head_dim elements with 16-byte loads;
// LPR lanes share a row (a wave instruction covers 64/LPR whole rows, 1 KB contiguous when the cache is
// dense), U rows per lane group are in flight.  Algorithmic bytes: B * kv_heads * kv * head_dim * esize.
#include "common.hpp"

namespace attwarp {

int launch_attn_step_dtype(int dtype, const void* attn, int nb, int heads, int64_t sb, int64_t sh, int64_t row_off,
                           int64_t skv, const int32_t* starts, int starts_mod, int max_start, int ntok, void* out,
                           hipStream_t st);

namespace probe {

constexpr int NT = 256;
constexpr int MAXCPL = 4;   // 16-byte chunks of a row per lane
constexpr int MAX_KV = 15872;

typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename T> struct Chunk { static constexpr int NE = 16 / sizeof(T); };

template <typename T> __device__ __forceinline__ void decode(const u4& v, double* d);
template <> __device__ __forceinline__ void decode<float>(const u4& v, double* d) {
  const f4 f = __builtin_bit_cast(f4, v);
#pragma unroll
  for (int i = 0; i < 4; ++i) d[i] = (double)f[i];
}
template <> __device__ __forceinline__ void decode<__half>(const u4& v, double* d) {
  const h8 h = __builtin_bit_cast(h8, v);
#pragma unroll
  for (int i = 0; i < 8; ++i) d[i] = (double)(float)h[i];
}
template <> __device__ __forceinline__ void decode<__hip_bfloat16>(const u4& v, double* d) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    d[2 * i] = (double)__uint_as_float(v[i] << 16);
    d[2 * i + 1] = (double)__uint_as_float(v[i] & 0xffff0000u);
  }
}

struct Params {
  int heads, group, kv, ntok;
  int64_t q_sb, q_sh, k_sb, k_sh, k_st;   // element strides
  const int32_t* kv_begin;                // [B] or null
  const int32_t* starts;                  // [B]
  float scale;
  int32_t* zero;                          // workspace word the chained reduction reads as its slice start
};

// grid = heads * B.  CPL = 16-byte chunks of a row per lane; U rows per lane group in flight (8 loads per lane).
template <typename T, int LPR, int CPL>
__global__ __launch_bounds__(NT) void attn_probe_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                        const Params p, T* __restrict__ probs) {
  extern __shared__ float logit[];                 // kv floats
  __shared__ double red[NT / WAVE];
  __shared__ float fred[NT / WAVE];
  constexpr int NE = Chunk<T>::NE;
  constexpr int G = NT / LPR;                      // lane groups = rows per sweep
  constexpr int U = CPL == 1 ? 8 : (CPL == 2 ? 4 : 2);
  // XCD-aware block order (blocks bid, bid+8, ... share an XCD and its L2): each XCD gets a contiguous range of
  // (sample, head) pairs, so the query heads of one key head run next to each other on ONE L2 and grouped-query
  // attention fetches a key row from HBM once per group instead of once per head.
  int bid = blockIdx.x;
  {
    const int n = gridDim.x, q = n >> 3, r = n & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r) ? xcd * (q + 1) + idx : r * (q + 1) + (xcd - r) * q + idx;
  }
  const int b = bid / p.heads, h = bid - b * p.heads;
  const int tid = threadIdx.x, g = tid / LPR, l = tid % LPR;
  const int kv = p.kv;
  const int kb = p.kv_begin ? min(max(p.kv_begin[b], 0), kv) : 0;
  if (h == 0 && b == 0 && tid == 0) *p.zero = 0;

  // this lane's slices of the query vector, as double
  double qd[CPL][NE];
  {
    const char* qp = reinterpret_cast<const char*>(q + (int64_t)b * p.q_sb + (int64_t)h * p.q_sh);
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const u4 v = *reinterpret_cast<const u4*>(qp + (size_t)(c * LPR + l) * 16);
      decode<T>(v, qd[c]);
    }
  }

  const char* kp = reinterpret_cast<const char*>(k + (int64_t)b * p.k_sb + (int64_t)(h / p.group) * p.k_sh) +
                   (size_t)l * 16;
  const int64_t rstride = p.k_st * (int64_t)sizeof(T);
  for (int j0 = kb + g; j0 < kv; j0 += G * U) {
    u4 v[U][CPL];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const char* rp = kp + (int64_t)min(j0 + u * G, kv - 1) * rstride;   // clamped: tail rows re-read, not stored
#pragma unroll
      for (int c = 0; c < CPL; ++c) v[u][c] = *reinterpret_cast<const u4*>(rp + (size_t)c * LPR * 16);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      double acc = 0.0;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        double kd[NE];
        decode<T>(v[u][c], kd);
#pragma unroll
        for (int i = 0; i < NE; ++i) acc = fma(qd[c][i], kd[i], acc);
      }
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, WAVE);
      const int j = j0 + u * G;
      if (l == 0 && j < kv) {
        const T w = from_f64<T>(acc);                                  // matmul output in the model dtype
        logit[j] = to_f32<T>(from_f32<T>(fmul(to_f32<T>(w), p.scale)));  // * scaling, rounded to the model dtype
      }
    }
  }
  __syncthreads();

  // softmax(dtype=float32) over the valid positions
  float m = -INFINITY;
  for (int j = kb + tid; j < kv; j += NT) m = fmaxf(m, logit[j]);
  m = wave_max(m);
  if ((tid & (WAVE - 1)) == 0) fred[tid / WAVE] = m;
  __syncthreads();
  m = fred[0];
  for (int w = 1; w < NT / WAVE; ++w) m = fmaxf(m, fred[w]);
  double s = 0.0;
  for (int j = kb + tid; j < kv; j += NT) {
    const float e = (float)exp((double)fsub(logit[j], m));
    logit[j] = e;
    s += (double)e;
  }
  const float den = (float)block_sum(s, red);      // (block_sum's barriers also publish the e values)
  const int st = min(max(p.starts[b], 0), p.kv - p.ntok);   // clamped like the hook slice (no out-of-row reads)
  T* pr = probs + ((int64_t)b * p.heads + h) * p.ntok;
  for (int t = tid; t < p.ntok; t += NT) {
    const int j = st + t;
    const float pv = (j >= kb && j < kv) ? logit[j] / den : 0.0f;
    pr[t] = from_f32<T>(pv);
  }
}

template <typename T, int LPR, int CPL>
static int launch_k(const void* q, const void* k, const Params& p, int B, void* probs, hipStream_t st) {
  hipLaunchKernelGGL((attn_probe_kernel<T, LPR, CPL>), dim3((unsigned)(p.heads * B)), dim3(NT), (size_t)p.kv * sizeof(float), st,
                     (const T*)q, (const T*)k, p, (T*)probs);
  return check_launch("attn_probe_kernel");
}

// chunks per row nc = LPR * CPL with LPR the largest power of two <= 16 dividing nc: below 16 lanes CPL is odd
template <typename T>
static int launch_t(const void* q, const void* k, const Params& p, int lpr, int cpl, int B, void* probs,
                    hipStream_t st) {
#define ATTWARP_PROBE_CASE(L, C) \
  if (lpr == L && cpl == C) return launch_k<T, L, C>(q, k, p, B, probs, st);
  ATTWARP_PROBE_CASE(16, 1) ATTWARP_PROBE_CASE(16, 2) ATTWARP_PROBE_CASE(16, 3) ATTWARP_PROBE_CASE(16, 4)
  ATTWARP_PROBE_CASE(8, 1) ATTWARP_PROBE_CASE(8, 3)
  ATTWARP_PROBE_CASE(4, 1) ATTWARP_PROBE_CASE(4, 3)
  ATTWARP_PROBE_CASE(2, 1) ATTWARP_PROBE_CASE(2, 3)
  ATTWARP_PROBE_CASE(1, 1) ATTWARP_PROBE_CASE(1, 3)
#undef ATTWARP_PROBE_CASE
  return fail(ATTWARP_E_UNSUPPORTED, "attn_probe_last_query: no kernel for %d lanes x %d chunks per lane", lpr, cpl);
}

static size_t esize(int dtype) { return dtype == ATTWARP_F32 ? 4 : 2; }

}  // namespace probe
}  // namespace attwarp

using namespace attwarp;

extern "C" size_t attwarp_attn_probe_workspace_bytes(int dtype, int B, int heads, int ntok) {
  if (B <= 0 || heads <= 0 || ntok <= 0) return 0;
  const size_t n = (size_t)B * heads * ntok * probe::esize(dtype);
  return 16 + ((n + 15) & ~(size_t)15);
}

extern "C" int attwarp_attn_probe_last_query(const void* q, const void* k, int dtype, int B, int heads, int kv_heads,
                                             int head_dim, int kv_len, int64_t q_stride_b, int64_t q_stride_h,
                                             int64_t k_stride_b, int64_t k_stride_h, int64_t k_stride_t,
                                             const int32_t* kv_begin, const int32_t* starts, int ntok, float scaling,
                                             void* out, void* ws, void* stream) {
  ATTWARP_REQUIRE(q && k && starts && out && ws, "attn_probe_last_query: null pointer");
  ATTWARP_REQUIRE(B > 0 && heads > 0 && kv_heads > 0 && head_dim > 0 && kv_len > 0 && ntok > 0,
                  "attn_probe_last_query: non-positive size");
  ATTWARP_REQUIRE(dtype == ATTWARP_F32 || dtype == ATTWARP_F16 || dtype == ATTWARP_BF16,
                  "attn_probe_last_query: dtype must be F32, F16 or BF16 (got %d)", dtype);
  ATTWARP_REQUIRE(heads % kv_heads == 0, "attn_probe_last_query: heads=%d is not a multiple of kv_heads=%d", heads,
                  kv_heads);
  ATTWARP_REQUIRE(ntok <= kv_len, "attn_probe_last_query: ntok=%d > kv_len=%d", ntok, kv_len);
  if ((long long)B * heads > 2147483647LL) return fail(ATTWARP_E_UNSUPPORTED, "attn_probe_last_query: B*heads too large");
  if (ntok > 1024) return fail(ATTWARP_E_UNSUPPORTED, "attn_probe_last_query: ntok=%d > 1024", ntok);
  if (kv_len > probe::MAX_KV)
    return fail(ATTWARP_E_UNSUPPORTED, "attn_probe_last_query: kv_len=%d > %d", kv_len, probe::MAX_KV);
  const size_t es = probe::esize(dtype);
  const int per = (int)(16 / es);                      // elements per 16-byte chunk
  if (head_dim % per != 0)
    return fail(ATTWARP_E_UNSUPPORTED, "attn_probe_last_query: head_dim=%d is not a multiple of %d", head_dim, per);
  const int64_t strides[5] = {q_stride_b, q_stride_h, k_stride_b, k_stride_h, k_stride_t};
  for (int i = 0; i < 5; ++i)
    if (strides[i] % per != 0)
      return fail(ATTWARP_E_UNSUPPORTED, "attn_probe_last_query: strides must be multiples of %d elements (16 bytes)",
                  per);
  if (((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(ws)) & 15u) != 0)
    return fail(ATTWARP_E_UNSUPPORTED, "attn_probe_last_query: q, k and ws must be 16-byte aligned");
  const int nc = head_dim / per;                       // chunks per row
  int lpr = 16;
  while (nc % lpr != 0) lpr >>= 1;
  const int cpl = nc / lpr;
  if (cpl > probe::MAXCPL)
    return fail(ATTWARP_E_UNSUPPORTED, "attn_probe_last_query: head_dim=%d needs %d chunks per lane (max %d)",
                head_dim, cpl, probe::MAXCPL);

  probe::Params p;
  p.heads = heads; p.group = heads / kv_heads; p.kv = kv_len; p.ntok = ntok;
  p.q_sb = q_stride_b; p.q_sh = q_stride_h; p.k_sb = k_stride_b; p.k_sh = k_stride_h; p.k_st = k_stride_t;
  p.kv_begin = kv_begin; p.starts = starts; p.scale = scaling;
  p.zero = reinterpret_cast<int32_t*>(ws);
  void* probs = reinterpret_cast<char*>(ws) + 16;
  hipStream_t st = as_stream(stream);
  int rc;
  switch (dtype) {
    case ATTWARP_F32: rc = probe::launch_t<float>(q, k, p, lpr, cpl, B, probs, st); break;
    case ATTWARP_F16: rc = probe::launch_t<__half>(q, k, p, lpr, cpl, B, probs, st); break;
    default: rc = probe::launch_t<__hip_bfloat16>(q, k, p, lpr, cpl, B, probs, st); break;
  }
  if (rc) return rc;
  // A1 on the probed rows: probs [B,heads,ntok], slice start 0 for every sample
  return launch_attn_step_dtype(dtype, probs, B, heads, (int64_t)heads * ntok, ntok, 0, 1, p.zero, 1, 0, ntok, out, st);
}
