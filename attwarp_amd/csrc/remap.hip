// A12 / A13 tail: bilinear resample with replicate border and SEPARABLE maps.
//
// Replaces cv2.remap(img, meshgrid(map_x, map_y), INTER_LINEAR, BORDER_REPLICATE)
// (reference: AGW/new_method.py:268-271, MN/checkpoint_utils.py:195-198).  The
// reference materialises dense [H_out,W_out] float32 maps; they are the outer
// product of two vectors, so the kernels take map_x[B,W_out], map_y[B,H_out].
//
// Arithmetic (mode EXACT, mirrored by oracle/warp_oracle.py::remap_bilinear):
//   i0 = floor(m), f = m - i0, taps clamped to the image (replicate border)
//   v  = a + fy*(c - a)       vertical lerp of the two source rows   (per tap column)
//   out= v0 + fx*(v1 - v0)    horizontal lerp
// every operation individually rounded to float32 (no FMA); uint8 sources are
// converted to float32, interpolated the same way and rounded half-to-even.
//
// Kernels of the resample stage
//   remap_rows_kernel (remap_rows.hip) - the HBM-roofline path (float32): one workgroup streams a block
//                         of consecutive output rows; source rows are read once with 16-byte
//                         coalesced loads, blended vertically in registers, staged in LDS and
//                         gathered horizontally from LDS.  See DESIGN.md "K7".
//   remap_rows_u8_kernel (remap_u8.hip) - the uint8 counterpart (main_batched chain).
//   remap_gather_kernel (this file) - generic fallback (any size / dtype / mode): one thread per output
//                         element, four global taps served by L1/L2.
#include "common.hpp"

namespace attwarp {

struct Taps {
  int i0, i1;
  float f;
};

__device__ __forceinline__ Taps taps_exact(float m, int size) {
  const float mc = clamp_coord(m, size);
  const float fl = floorf(mc);
  Taps t;
  t.f = fsub(mc, fl);
  const int i = (int)fl;
  t.i0 = min(max(i, 0), size - 1);
  t.i1 = min(max(i + 1, 0), size - 1);
  return t;
}

// OpenCV's INTER_BITS=5 coordinate quantisation: q = cvRound(m*32), index q>>5, fraction q&31.
__device__ __forceinline__ Taps taps_cv2(float m, int size) {
  const int q = cv_round_q5(m);
  const int i = q >> 5;
  Taps t;
  t.f = (float)(q & 31) * 0.03125f;
  t.i0 = min(max(i, 0), size - 1);
  t.i1 = min(max(i + 1, 0), size - 1);
  return t;
}

template <int MODE>
__device__ __forceinline__ Taps taps(float m, int size) {
  return MODE == ATTWARP_CV2 ? taps_cv2(m, size) : taps_exact(m, size);
}

template <typename T, int MODE>
__device__ __forceinline__ T blend(T p00, T p01, T p10, T p11, float fx, float fy);

template <>
__device__ __forceinline__ float blend<float, ATTWARP_EXACT>(float p00, float p01, float p10, float p11, float fx,
                                                             float fy) {
  const float v0 = lerp_rn(p00, p10, fy);
  const float v1 = lerp_rn(p01, p11, fy);
  return lerp_rn(v0, v1, fx);
}
template <>
__device__ __forceinline__ uint8_t blend<uint8_t, ATTWARP_EXACT>(uint8_t p00, uint8_t p01, uint8_t p10, uint8_t p11,
                                                                 float fx, float fy) {
  const float v = blend<float, ATTWARP_EXACT>((float)p00, (float)p01, (float)p10, (float)p11, fx, fy);
  return (uint8_t)fminf(fmaxf(rintf(v), 0.0f), 255.0f);
}
// OpenCV float path: ((p00*w00 + p01*w01) + p10*w10) + p11*w11 with table weights
// w = (1-ty|ty)*(1-tx|tx); t = k/32 so the weights are exact in float32.
template <>
__device__ __forceinline__ float blend<float, ATTWARP_CV2>(float p00, float p01, float p10, float p11, float fx,
                                                           float fy) {
  const float ox = fsub(1.0f, fx), oy = fsub(1.0f, fy);
  const float w00 = fmul(oy, ox), w01 = fmul(oy, fx), w10 = fmul(fy, ox), w11 = fmul(fy, fx);
  return fadd(fadd(fadd(fmul(p00, w00), fmul(p01, w01)), fmul(p10, w10)), fmul(p11, w11));
}
// OpenCV uint8 path: int16 weights = saturate_cast<short>(w * 2^15), result (sum + 2^14) >> 15.
template <>
__device__ __forceinline__ uint8_t blend<uint8_t, ATTWARP_CV2>(uint8_t p00, uint8_t p01, uint8_t p10, uint8_t p11,
                                                               float fx, float fy) {
  const int kx = (int)(fx * 32.0f), ky = (int)(fy * 32.0f);  // exact: fx = k/32
  const int w00 = min((32 - ky) * (32 - kx) * 32, 32767), w01 = (32 - ky) * kx * 32;
  const int w10 = ky * (32 - kx) * 32, w11 = ky * kx * 32;
  const int acc = (int)p00 * w00 + (int)p01 * w01 + (int)p10 * w10 + (int)p11 * w11;
  return (uint8_t)min(max((acc + (1 << 14)) >> 15, 0), 255);
}

// ------------------------------------------------------------------------------
// Generic gather kernel.  grid = (ceil(Wo*CS / 256), Ho, B)
//   HWC: one thread per interleaved output element e = x*C + c        (CS = C, planes = 1)
//   CHW: one thread per output column x, loops the C planes           (CS = 1, planes = C)
// ------------------------------------------------------------------------------
template <typename T, int MODE, int LAYOUT>
__global__ __launch_bounds__(256) void remap_gather_kernel(const T* __restrict__ src, T* __restrict__ dst, int C,
                                                           int H, int W, int Ho, int Wo,
                                                           const float* __restrict__ map_x,
                                                           const float* __restrict__ map_y) {
  const int b = blockIdx.z, y = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int CS = (LAYOUT == ATTWARP_HWC) ? C : 1;
  if (e >= Wo * CS) return;
  const int x = (LAYOUT == ATTWARP_HWC) ? e / C : e;
  const int c0 = (LAYOUT == ATTWARP_HWC) ? e - x * C : 0;
  const Taps tx = taps<MODE>(map_x[(size_t)b * Wo + x], W);
  const Taps ty = taps<MODE>(map_y[(size_t)b * Ho + y], H);
  if (LAYOUT == ATTWARP_HWC) {
    const T* s = src + (size_t)b * H * W * C;
    const size_t r0 = (size_t)ty.i0 * W * C, r1 = (size_t)ty.i1 * W * C;
    const int a0 = tx.i0 * C + c0, a1 = tx.i1 * C + c0;
    const T v = blend<T, MODE>(s[r0 + a0], s[r0 + a1], s[r1 + a0], s[r1 + a1], tx.f, ty.f);
    dst[((size_t)b * Ho + y) * Wo * C + e] = v;
  } else {
    for (int c = 0; c < C; ++c) {
      const T* s = src + ((size_t)b * C + c) * H * W;
      const size_t r0 = (size_t)ty.i0 * W, r1 = (size_t)ty.i1 * W;
      const T v = blend<T, MODE>(s[r0 + tx.i0], s[r0 + tx.i1], s[r1 + tx.i0], s[r1 + tx.i1], tx.f, ty.f);
      dst[(((size_t)b * C + c) * Ho + y) * Wo + x] = v;
    }
  }
}

template <typename T, int MODE, int LAYOUT>
static int launch_gather(const void* src, void* dst, int B, int C, int H, int W, int Ho, int Wo, const float* mx,
                         const float* my, hipStream_t st) {
  const int CS = (LAYOUT == ATTWARP_HWC) ? C : 1;
  dim3 grid((Wo * CS + 255) / 256, Ho, B);
  hipLaunchKernelGGL((remap_gather_kernel<T, MODE, LAYOUT>), grid, dim3(256), 0, st, (const T*)src, (T*)dst, C, H, W,
                     Ho, Wo, mx, my);
  return check_launch("remap_gather_kernel");
}

struct StepExtra;
int launch_remap_rows(const float* src, float* dst, int layout, int B, int C, int H, int W, int Ho, int Wo,
                      const float* mx, const float* my, int mode, hipStream_t st, bool* handled, StepExtra* ex);
int launch_remap_rows_u8(const uint8_t* src, uint8_t* dst, int layout, int B, int C, int H, int W, int Ho, int Wo,
                         const float* mx, const float* my, int mode, hipStream_t st, bool* handled);

}  // namespace attwarp

using namespace attwarp;

extern "C" int attwarp_remap_bilinear(const void* src, void* dst, int dtype, int layout, int B, int C, int H, int W,
                                      int H_out, int W_out, const float* map_x, const float* map_y, int mode,
                                      void* stream) {
  ATTWARP_REQUIRE(src && dst && map_x && map_y, "remap_bilinear: null pointer");
  ATTWARP_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && H_out > 0 && W_out > 0, "remap_bilinear: non-positive size");
  ATTWARP_REQUIRE(dtype == ATTWARP_F32 || dtype == ATTWARP_U8, "remap_bilinear: dtype must be F32 or U8 (got %d)", dtype);
  ATTWARP_REQUIRE(layout == ATTWARP_HWC || layout == ATTWARP_CHW, "remap_bilinear: unknown layout %d", layout);
  ATTWARP_REQUIRE(mode == ATTWARP_EXACT || mode == ATTWARP_CV2, "remap_bilinear: unknown mode %d", mode);
  if (C > 4) return fail(ATTWARP_E_UNSUPPORTED, "remap_bilinear: C=%d > 4", C);
  if (B > 65535 || H_out > 65535) return fail(ATTWARP_E_UNSUPPORTED, "remap_bilinear: B and H_out must be <= 65535");
  if ((long long)W_out * C > 2147483647LL / 2 || (long long)H * W * C > 2147483647LL)
    return fail(ATTWARP_E_UNSUPPORTED, "remap_bilinear: image too large");
  hipStream_t st = as_stream(stream);

  if (dtype == ATTWARP_F32) {
    bool handled = false;
    int rc = launch_remap_rows((const float*)src, (float*)dst, layout, B, C, H, W, H_out, W_out, map_x, map_y, mode, st,
                               &handled, nullptr);
    if (handled) return rc;
  } else {
    bool handled = false;
    int rc = launch_remap_rows_u8((const uint8_t*)src, (uint8_t*)dst, layout, B, C, H, W, H_out, W_out, map_x, map_y,
                                  mode, st, &handled);
    if (handled) return rc;
  }

#define ATTWARP_DISPATCH(T)                                                                                  \
  if (mode == ATTWARP_EXACT && layout == ATTWARP_HWC)                                                        \
    return launch_gather<T, ATTWARP_EXACT, ATTWARP_HWC>(src, dst, B, C, H, W, H_out, W_out, map_x, map_y, st); \
  if (mode == ATTWARP_EXACT && layout == ATTWARP_CHW)                                                        \
    return launch_gather<T, ATTWARP_EXACT, ATTWARP_CHW>(src, dst, B, C, H, W, H_out, W_out, map_x, map_y, st); \
  if (mode == ATTWARP_CV2 && layout == ATTWARP_HWC)                                                          \
    return launch_gather<T, ATTWARP_CV2, ATTWARP_HWC>(src, dst, B, C, H, W, H_out, W_out, map_x, map_y, st);   \
  return launch_gather<T, ATTWARP_CV2, ATTWARP_CHW>(src, dst, B, C, H, W, H_out, W_out, map_x, map_y, st);

  if (dtype == ATTWARP_F32) {
    ATTWARP_DISPATCH(float)
  } else {
    ATTWARP_DISPATCH(uint8_t)
  }
#undef ATTWARP_DISPATCH
}
