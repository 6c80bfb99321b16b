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

// float64 images (warp_from_cdf_torch hands the image to cv2.remap in its own dtype, MN/checkpoint_utils.py:152,195):
// OpenCV's CV_64F path keeps the float32 table weights and accumulates the four products in double, left to right;
// exact mode: the three lerps in double with the float32 coordinate fractions.
template <>
__device__ __forceinline__ double blend<double, ATTWARP_EXACT>(double p00, double p01, double p10, double p11, float fx,
                                                               float fy) {
  const double v0 = __dadd_rn(p00, __dmul_rn((double)fy, __dsub_rn(p10, p00)));
  const double v1 = __dadd_rn(p01, __dmul_rn((double)fy, __dsub_rn(p11, p01)));
  return __dadd_rn(v0, __dmul_rn((double)fx, __dsub_rn(v1, v0)));
}
template <>
__device__ __forceinline__ double blend<double, ATTWARP_CV2>(double p00, double p01, double p10, double p11, float fx,
                                                             float fy) {
  const float ox = fsub(1.0f, fx), oy = fsub(1.0f, fy);
  const double w00 = fmul(oy, ox), w01 = fmul(oy, fx), w10 = fmul(fy, ox), w11 = fmul(fy, fx);
  return __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(p00, w00), __dmul_rn(p01, w01)), __dmul_rn(p10, w10)), __dmul_rn(p11, w11));
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

// ------------------------------------------------------------------------------
// cv2.resize(image, (W_out, H_out), interpolation=INTER_LINEAR)  (AGW/new_method.py:369, reached from :478 when the
// attention map's size differs from the image's -- a no-op in both reference drivers).  OpenCV's published algorithm
// (modules/imgproc/src/resize.cpp), parity unpinned like cv2.remap:
//   fx = float((dx + 0.5) * scale_x - 0.5), sx = floor(fx), fx -= sx; sx < 0 -> (0, fx = 0); sx >= W-1 -> (W-1, fx = 0)
//   rows likewise without the fx reset, indices clipped to the image;
//   uint8: coefficients saturate_cast<short>(c * 2048); horizontal pass D = S[sx]*a0 + S[sx+1]*a1 (int32), vertical
//          uchar((((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2);
//   float32: D = S[sx]*a0 + S[sx+1]*a1, out = D0*b0 + D1*b1 (each operation rounded);
//   an exact 2 x 2 decimation is INTER_AREA: (s00 + s01 + s10 + s11 + 2) >> 2 (float: sum * 0.25).
// One thread per output element of an interleaved [B,H,W,C] image; a gather kernel, this is not a hot path.
struct ResizeAxis {
  int s0, s1;
  float f;
};
__device__ __forceinline__ ResizeAxis resize_axis(int d, double scale, int size, bool reset_f) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f = fsub(f, (float)s);
  if (reset_f) {
    if (s < 0) { f = 0.0f; s = 0; }
    if (s >= size - 1) { f = 0.0f; s = size - 1; }
  }
  ResizeAxis a;
  a.f = f;
  a.s0 = min(max(s, 0), size - 1);
  a.s1 = min(max(s + 1, 0), size - 1);
  return a;
}
template <typename T>
__global__ __launch_bounds__(256) void resize_linear_kernel(const T* __restrict__ src, T* __restrict__ dst, int C, int H,
                                                            int W, int Ho, int Wo, double scale_x, double scale_y,
                                                            int area2) {
  const int b = blockIdx.z, y = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= Wo * C) return;
  const int x = e / C, c = e - x * C;
  const T* s = src + (size_t)b * H * W * C;
  T* o = dst + ((size_t)b * Ho + y) * Wo * C + e;
  if (area2) {
    const T* p = s + ((size_t)(2 * y) * W + 2 * x) * C + c;
    if (sizeof(T) == 1) {
      *o = (T)(((int)p[0] + (int)p[C] + (int)p[(size_t)W * C] + (int)p[(size_t)W * C + C] + 2) >> 2);
    } else {
      *o = (T)fmul(fadd(fadd(fadd((float)p[0], (float)p[C]), (float)p[(size_t)W * C]), (float)p[(size_t)W * C + C]), 0.25f);
    }
    return;
  }
  const ResizeAxis ax = resize_axis(x, scale_x, W, true), ay = resize_axis(y, scale_y, H, false);
  const T* r0 = s + (size_t)ay.s0 * W * C;
  const T* r1 = s + (size_t)ay.s1 * W * C;
  const int i0 = ax.s0 * C + c, i1 = ax.s1 * C + c;
  if (sizeof(T) == 1) {
    const int a0 = __float2int_rn(fmul(fsub(1.0f, ax.f), 2048.0f)), a1 = __float2int_rn(fmul(ax.f, 2048.0f));
    const int b0 = __float2int_rn(fmul(fsub(1.0f, ay.f), 2048.0f)), b1 = __float2int_rn(fmul(ay.f, 2048.0f));
    const int d0 = (int)r0[i0] * a0 + (int)r0[i1] * a1, d1 = (int)r1[i0] * a0 + (int)r1[i1] * a1;
    const int v = (((b0 * (d0 >> 4)) >> 16) + ((b1 * (d1 >> 4)) >> 16) + 2) >> 2;
    *o = (T)min(max(v, 0), 255);
  } else {
    const float a0 = fsub(1.0f, ax.f), a1 = ax.f, b0 = fsub(1.0f, ay.f), b1 = ay.f;
    const float d0 = fadd(fmul((float)r0[i0], a0), fmul((float)r0[i1], a1));
    const float d1 = fadd(fmul((float)r1[i0], a0), fmul((float)r1[i1], a1));
    *o = (T)fadd(fmul(d0, b0), fmul(d1, b1));
  }
}

}  // namespace attwarp

using namespace attwarp;

extern "C" int attwarp_resize_linear(const void* src, void* dst, int dtype, int B, int C, int H, int W, int H_out,
                                     int W_out, void* stream) {
  ATTWARP_REQUIRE(src && dst, "resize_linear: null pointer");
  ATTWARP_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && H_out > 0 && W_out > 0, "resize_linear: non-positive size");
  ATTWARP_REQUIRE(dtype == ATTWARP_F32 || dtype == ATTWARP_U8, "resize_linear: dtype must be F32 or U8 (got %d)", dtype);
  if (B > 65535 || H_out > 65535 || (long long)W_out * C > 2147483647LL / 2 || (long long)H * W * C > 2147483647LL)
    return fail(ATTWARP_E_UNSUPPORTED, "resize_linear: image too large");
  const double sx = 1.0 / ((double)W_out / (double)W), sy = 1.0 / ((double)H_out / (double)H);   // 1 / inv_scale, as OpenCV
  const int area2 = (W == 2 * W_out && H == 2 * H_out) ? 1 : 0;
  dim3 grid((W_out * C + 255) / 256, H_out, B);
  if (dtype == ATTWARP_F32)
    hipLaunchKernelGGL(resize_linear_kernel<float>, grid, dim3(256), 0, as_stream(stream), (const float*)src,
                       (float*)dst, C, H, W, H_out, W_out, sx, sy, area2);
  else
    hipLaunchKernelGGL(resize_linear_kernel<uint8_t>, grid, dim3(256), 0, as_stream(stream), (const uint8_t*)src,
                       (uint8_t*)dst, C, H, W, H_out, W_out, sx, sy, area2);
  return check_launch("resize_linear_kernel");
}

extern "C" int attwarp_remap_bilinear(const void* src, void* dst, int dtype, int layout, int B, int C, int H, int W,
                                      int H_out, int W_out, const float* map_x, const float* map_y, int mode,
                                      void* stream) {
  ATTWARP_REQUIRE(src && dst && map_x && map_y, "remap_bilinear: null pointer");
  ATTWARP_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && H_out > 0 && W_out > 0, "remap_bilinear: non-positive size");
  ATTWARP_REQUIRE(dtype == ATTWARP_F32 || dtype == ATTWARP_U8 || dtype == ATTWARP_F64,
                  "remap_bilinear: dtype must be F32, U8 or F64 (got %d)", dtype);
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
  } else if (dtype == ATTWARP_U8) {
    bool handled = false;
    int rc = launch_remap_rows_u8((const uint8_t*)src, (uint8_t*)dst, layout, B, C, H, W, H_out, W_out, map_x, map_y,
                                  mode, st, &handled);
    if (handled) return rc;
  }

  // (tuning flavour, remap_variant = 3: the tests' proof of WHICH kernel served a shape -- refuse instead of falling back)
  if (tune(TUNE_REMAP_VARIANT) == 3) return fail(ATTWARP_E_UNSUPPORTED, "remap_bilinear: this request takes the generic gather kernel");

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
  } else if (dtype == ATTWARP_F64) {
    ATTWARP_DISPATCH(double)
  } else {
    ATTWARP_DISPATCH(uint8_t)
  }
#undef ATTWARP_DISPATCH
}
