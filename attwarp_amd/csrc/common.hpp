// Shared host/device helpers for libattwarp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/attwarp.h"

namespace attwarp {

constexpr int WAVE = 64;  // CDNA wavefront width

// ---- thread-local error text -------------------------------------------------
char* error_buffer();
int fail(int code, const char* fmt, ...);

#define ATTWARP_REQUIRE(cond, ...)                                   \
  do {                                                               \
    if (!(cond)) return ::attwarp::fail(ATTWARP_E_ARG, __VA_ARGS__); \
  } while (0)

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ATTWARP_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return ATTWARP_OK;
}

// ---- float arithmetic with one rounding per operation -------------------------
// The oracle defines every float32 stage as a sequence of individually rounded
// operations; the library is built with -ffp-contract=off and these wrappers make
// the intent explicit where bit parity depends on it.
__device__ __forceinline__ float fmul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float fadd(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float fsub(float a, float b) { return __fsub_rn(a, b); }
// a + t*(b-a), three roundings (no FMA): the oracle's _lerp
// Coordinate -> taps conventions shared by every resample kernel (and restated by oracle/):
//  * cv2 mode: q = cvRound(32 * m) as OpenCV computes it on x86 (cvtss2si / cvtps2dq): a product that is NaN or does
//    not fit a 32-bit integer becomes INT_MIN ("integer indefinite"), i.e. pixel 0 with a zero fraction after the
//    replicate border -- for NaN, +-Inf and huge coordinates of EITHER sign;
//  * exact mode: the coordinate is clamped to [-1, size] first (NaN counts as -1), so non-finite coordinates land on an
//    edge pixel and never produce a NaN weight.
__device__ __forceinline__ int cv_round_q5(float m) {
  const float s = fmul(m, 32.0f);
  return (s >= -2147483648.0f && s < 2147483648.0f) ? __float2int_rn(s) : (int)0x80000000;
}
__device__ __forceinline__ float clamp_coord(float m, int size) { return fminf(fmaxf(m, -1.0f), (float)size); }
__device__ __forceinline__ float lerp_rn(float a, float b, float t) { return fadd(a, fmul(t, fsub(b, a))); }

// ---- wave / block reductions ---------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}
// The same butterfly (lane l adds the value of lane l ^ o, o = 32, 16, 8, 4, 2, 1: identical bits, float addition is
// commutative) without the six LDS round trips of ds_bpermute: v_permlane32_swap / v_permlane16_swap (gfx950) for
// o = 32 / 16, DPP row_ror:8 for o = 8, two bank-masked row shifts for o = 4, quad permutes for o = 2, 1.
template <int CTRL, int BANKS>
__device__ __forceinline__ float dpp_f32(float old, float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(old), __float_as_uint(v), CTRL, 0xf, BANKS, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);                    // o = 32
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(b[0]) + __uint_as_float(b[1]);                    // o = 16
  v = dpp_f32<0x128, 0xf>(0.0f, v) + v;                                 // o = 8: row_ror:8
  float t = dpp_f32<0x104, 0x5>(0.0f, v);                               // o = 4: lanes 0-3, 8-11 of a row read lane + 4 (row_shl:4)
  t = dpp_f32<0x114, 0xa>(t, v);                                        //        lanes 4-7, 12-15 read lane - 4 (row_shr:4)
  v = t + v;
  v = dpp_f32<0x4e, 0xf>(0.0f, v) + v;                                  // o = 2: quad_perm [2,3,0,1]
  v = dpp_f32<0xb1, 0xf>(0.0f, v) + v;                                  // o = 1: quad_perm [1,0,3,2]
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = WAVE / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = WAVE / 2; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, WAVE));
  return v;
}

// Block-wide sum in double. `scratch` needs blockDim.x/64 doubles. All threads get the result.
// The order is fixed (lane tree, then waves ascending) so results are run-to-run identical.
__device__ __forceinline__ double block_sum(double v, double* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE, nw = (blockDim.x + WAVE - 1) / WAVE;
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < nw; ++i) t += scratch[i];
  return t;
}

// ---- element loads as float / double -------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<__half>(__half v) { return __half2float(v); }
template <> __device__ __forceinline__ float to_f32<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }
template <> __device__ __forceinline__ float to_f32<uint8_t>(uint8_t v) { return (float)v; }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __half from_f32<__half>(float v) { return __float2half_rn(v); }
template <> __device__ __forceinline__ __hip_bfloat16 from_f32<__hip_bfloat16>(float v) { return __float2bfloat16(v); }

template <typename T> __device__ __forceinline__ T from_f64(double v);
template <> __device__ __forceinline__ float from_f64<float>(double v) { return (float)v; }
// double -> half/bf16 with a SINGLE rounding: round-to-odd into float first
// (truncate towards zero, set the sticky LSB), then round-to-nearest-even.
__device__ __forceinline__ float f64_to_f32_round_odd(double v) {
  float f = (float)v;
  if ((double)f != v) {
    uint32_t u = __float_as_uint(f);
    if (fabs((double)f) > fabs(v)) u -= 1u;  // magnitude bits: one step towards zero
    u |= 1u;
    f = __uint_as_float(u);
  }
  return f;
}
template <> __device__ __forceinline__ __half from_f64<__half>(double v) {
  return __float2half_rn(f64_to_f32_round_odd(v));
}
template <> __device__ __forceinline__ __hip_bfloat16 from_f64<__hip_bfloat16>(double v) {
  return __float2bfloat16(f64_to_f32_round_odd(v));
}

template <typename T> __device__ __forceinline__ T div_t(T a, T b) {   // a / b evaluated in dtype T
  return from_f32<T>(to_f32<T>(a) / to_f32<T>(b));
}

// A dword at ANY byte address: gfx950 runs with unaligned access mode on (one global / buffer dword instruction either
// way); the type only stops the compiler from assuming 4-byte alignment of rows like 683 x 3 bytes.
typedef uint32_t u32_una __attribute__((aligned(1)));

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// A launch gets at most 64 KB of dynamic LDS unless the kernel has been granted more (a CU of gfx950 has 160 KB).  The
// per-axis kernels keep a whole axis in LDS (12 bytes per pixel of max(W, H): 64 KB is reached near 4800 pixels), so
// their launchers call this first; it is idempotent and touches only the runtime's attribute of that one kernel.
constexpr size_t LDS_DEFAULT_MAX = 64 * 1024, LDS_CU_BYTES = 160 * 1024;
template <typename K>
inline int grant_dynamic_lds(K kernel, size_t bytes, const char* what) {
  if (bytes <= LDS_DEFAULT_MAX) return ATTWARP_OK;
  if (bytes > LDS_CU_BYTES) return fail(ATTWARP_E_UNSUPPORTED, "%s: needs %zu bytes of LDS (> %zu)", what, bytes, LDS_CU_BYTES);
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return fail(ATTWARP_E_UNSUPPORTED, "%s: %zu bytes of dynamic LDS refused: %s", what, bytes, hipGetErrorString(e));
  return ATTWARP_OK;
}

// ---- test / measurement overrides -----------------------------------------------
// Kernel selection is automatic.  The A/B tools and the parity tests need to force a particular variant
// (e.g. "the generic gather kernel") to check the variants against each other.  That exists ONLY in the tuning flavour
// of the library (libattwarp_hip_tuning.so, built with -DATTWARP_TUNING): attwarp_debug_set() stores into a table of
// relaxed atomics that tune() reads, -1 = automatic.  In the product library tune() is the constant -1, the symbol
// does not exist and every `tune(...)` branch folds away.  No environment variable is ever read by either.
enum TuneKey {
  TUNE_REMAP_VARIANT = 0,   // 1: generic gather kernel only; 2: uint8 cv2 on the float-pipeline rows kernel (not the integer form); 3: staged kernels only (a request that would take the gather kernel is refused)
  TUNE_REMAP_ROWS,          // output rows per workgroup (1..64)
  TUNE_REMAP_CHW_SPLIT,     // 0 / 1: planar images plane by plane
  TUNE_REMAP_TILED,         // 0: rows wider than the LDS row take the generic kernel
  TUNE_REMAP_TILE_KO,       // 8 / 12
  TUNE_REMAP_ALT,           // 0: every row block sweeps top-down
  TUNE_REMAP_NOSWZ,         // 0: contiguous range of row blocks per XCD, 1: plain order, g >= 2: XCDs interleaved in groups of g
  TUNE_REMAP_LDSPAD,        // extra dynamic LDS bytes (occupancy experiments)
  TUNE_REMAP_NT,            // 0 / 1: nontemporal loads for block-private source rows
  TUNE_LANCZOS_VARIANT,     // 1: two-kernel form, 2: row-block fused kernel (default: column-strip kernel when it applies)
  TUNE_LANCZOS_ROWS,
  TUNE_CLIP_VARIANT,        // 1: generic kernels
  TUNE_PROFILES_VARIANT,    // 1: generic (non byte-packed) profile kernel for uint8 attention
  TUNE_REMAP_CPW,           // row blocks per workgroup of the float32 staged resample (strided inside the image)
  TUNE_REMAP_SKEW,          // XCD x starts x * skew blocks into its contiguous range of row blocks
  TUNE_ATTN_HU,             // heads in flight per wave of the four-tokens-per-lane attention reduce (1, 2, 4, 8)
  TUNE_U8_AHEAD,            // integer uint8 resample: output rows whose source rows are requested ahead (1, 2, 4)
  TUNE_CHAIN_SEQ,           // block order of the one-launch mask-chain step: 0 interleaved, 1 P / L / R ranges, 2 P first then L / R interleaved, 3 P + R then L, 4 P + L then R, 10..99 P spread over that per cent of L / R
  TUNE_CHAIN_WAVES,         // 6 / 8: waves per SIMD its register allocation leaves room for
  TUNE_REMAP_CV2_DOUBLE,    // cv2 rows of 8-12 KB: 0 = one [top | bottom] LDS buffer and two barriers per row instead of two buffers and one (48 KB)
  TUNE_STEP_PRIO,           // one-launch steps: 1 = the latency-chain blocks (maps / finalize / revise) run at raised wave priority
  TUNE_BOUND,               // UPPER-BOUND experiments (outputs are garbage; docs/experiments.md round 5): bit 0 float32 cv2 resample stages
                            //   only the top row in a 3-row LDS pool (what a three-slot ring could save at most); bit 1 chain step: the
                            //   up-sampled masks of all images alias two images (L's stores and P's loads stay in L2: what fusing L
                            //   into P could save at most); bit 2 finalize body without its np.cumsum chain; bit 3 finalize body returns at once
  TUNE_TRACE_LO,            // block timeline of the one-launch steps (tools/gantt.py): bits 0..23 and 24..47 of the address of a
  TUNE_TRACE_HI,            //   device buffer of TRACE_WORDS x uint64 per block (layout below)
  TUNE_COUNT
};
#ifdef ATTWARP_TUNING
int tune(TuneKey k);        // current override or -1
#else
constexpr int tune(TuneKey) { return -1; }
#endif

#ifdef ATTWARP_TUNING
// Block timeline of the kernels that carry it, tuning flavour only (the product kernels carry none of it).  One record
// of TRACE_WORDS x uint64 per block (tools/gantt.py):
//   0 start, 1 end (s_memrealtime, 100 MHz)   2 kind   3 HW_ID | XCC_ID << 32   4 start, 5 end (s_memtime, shader clock)
//   6 what words 7.. hold: 1 = s_memtime stamps of the body's marks (zero = unused), 2 = cycle sums [stage, sync, gather, rows]
constexpr int TRACE_WORDS = 16;
inline unsigned long long* trace_buffer() {
  const int lo = tune(TUNE_TRACE_LO), hi = tune(TUNE_TRACE_HI);
  if (lo < 0 || hi < 0) return nullptr;
  return reinterpret_cast<unsigned long long*>(((unsigned long long)hi << 24) | (unsigned long long)lo);
}
#ifdef __HIPCC__
struct TraceStart { unsigned long long real, cyc; };
__device__ __forceinline__ TraceStart trace_now() { return TraceStart{__builtin_amdgcn_s_memrealtime(), __builtin_amdgcn_s_memtime()}; }
// a body's mark i (0..8): thread 0 stamps the shader clock
__device__ __forceinline__ void trace_mark(unsigned long long* tr, int i) {
  if (tr && threadIdx.x == 0) {
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long* r = tr + TRACE_WORDS * (size_t)blockIdx.x;
    r[6] = 1;
    r[7 + i] = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
  }
}
// after the block's work: every wave has passed the barrier, thread 0 writes the record
__device__ __forceinline__ void trace_block(unsigned long long* tr, TraceStart t0, int kind) {
  if (!tr) return;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long* r = tr + TRACE_WORDS * (size_t)blockIdx.x;
    r[0] = t0.real;
    r[1] = __builtin_amdgcn_s_memrealtime();
    r[2] = (unsigned long long)(long long)kind;
    r[3] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |             // HW_REG_HW_ID
           ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);      // HW_REG_XCC_ID
    r[4] = t0.cyc;
    r[5] = __builtin_amdgcn_s_memtime();
  }
}
#endif
#endif

}  // namespace attwarp
