/*
 * warp_ref.c -- plain-C CPU restatement of the AttWarp warp hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  It restates, scalar and single threaded, the same arithmetic as
 * oracle/warp_oracle.py (which is pinned to golden vectors captured from the reference) so the
 * reference's CPU path can be timed on the GPU box, where neither the reference's Python nor
 * OpenCV exist.  tests/test_oracle_c.py checks it bit-for-bit against the numpy oracle.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile): every float operation is
 * individually rounded, like the oracle and the HIP kernels.
 *
 * Reference lines followed (AGW = "Attention Guided Warping", MN = "model/marginalnet_full_dataset"):
 *   attn_reduce_stack   AGW/attention_extraction/llava.py:385-411
 *   marginals24         MN/checkpoint_utils.py:43-51
 *   right_inverse       MN/checkpoint_utils.py:64-131
 *   cdf_from_density    MN/checkpoint_utils.py:30-41
 *   axis_map_from_cdf   MN/checkpoint_utils.py:167-193 (np.interp restated from numpy's compiled_base.c)
 *   remap_bilinear      AGW/new_method.py:268-271, MN/checkpoint_utils.py:195-198 (cv2.remap stand-in,
 *                       exact bilinear, replicate border -- parity unpinned, see warp_oracle.py)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define API __attribute__((visibility("default")))

/* ---- A1 + A2 (float32) ------------------------------------------------------------------
 * A1 accumulates in float32 in the path's fixed order (oracle/warp_oracle.py::_row_sums_f32_tree, _head_sums_f32):
 * row sum = 64 lane sums (lane = (t mod 256) / 4, quads (x0 + x1) + (x2 + x3), 256-token blocks in order) combined by
 * the butterfly s[l] += s[l ^ o], o = 32 .. 1; head sum = ((0 + P_0) + P_1) + P_2) + P_3, P_w over heads w, w+4, ...
 * A2 (mean over steps) accumulates in double and rounds once. */
static float row_sum_tree(const float* r, int ntok) {
  float s[64], t[64];
  for (int l = 0; l < 64; ++l) s[l] = 0.0f;
  for (int blk = 0; blk * 256 < ntok; ++blk)
    for (int l = 0; l < 64; ++l) {
      const int t0 = blk * 256 + 4 * l;
      const float x0 = t0 < ntok ? r[t0] : 0.0f, x1 = t0 + 1 < ntok ? r[t0 + 1] : 0.0f;
      const float x2 = t0 + 2 < ntok ? r[t0 + 2] : 0.0f, x3 = t0 + 3 < ntok ? r[t0 + 3] : 0.0f;
      const float a = x0 + x1, b = x2 + x3;
      const float q = a + b;
      s[l] = s[l] + q;
    }
  for (int o = 32; o >= 1; o >>= 1) {
    for (int l = 0; l < 64; ++l) t[l] = s[l] + s[l ^ o];
    memcpy(s, t, sizeof(s));
  }
  return s[0];
}

API void oracle_attn_reduce_stack_f32(const float* rows, int T, int B, int heads, int kv, const int32_t* starts,
                                      int ntok, float* out /* [B,ntok] */) {
  double* acc_steps = (double*)calloc((size_t)ntok, sizeof(double));
  float* pw = (float*)calloc((size_t)4 * ntok, sizeof(float));
  for (int b = 0; b < B; ++b) {
    memset(acc_steps, 0, sizeof(double) * ntok);
    for (int t = 0; t < T; ++t) {
      memset(pw, 0, sizeof(float) * 4 * ntok);
      for (int h = 0; h < heads; ++h) {
        const float* r = rows + (((size_t)t * B + b) * heads + h) * kv + starts[b];
        const float den = row_sum_tree(r, ntok) + 1e-12f;
        float* p = pw + (size_t)(h & 3) * ntok;
        for (int i = 0; i < ntok; ++i) { const float q = r[i] / den; p[i] = p[i] + q; }
      }
      for (int i = 0; i < ntok; ++i) {
        float m = 0.0f;
        for (int w = 0; w < 4; ++w) m = m + pw[(size_t)w * ntok + i];
        acc_steps[i] += (double)(m / (float)heads);
      }
    }
    for (int i = 0; i < ntok; ++i) out[(size_t)b * ntok + i] = (float)acc_steps[i] / (float)T;
  }
  free(acc_steps);
  free(pw);
}

/* ---- A6 on an [n,n] map: px over columns, py over rows ----------------------------------- */
API void oracle_marginals(const float* A, int H, int W, float* px, float* py) {
  double tx = 0.0, ty = 0.0;
  for (int x = 0; x < W; ++x) {
    double s = 0.0;
    for (int y = 0; y < H; ++y) { float v = A[(size_t)y * W + x]; s += (double)(v != v ? v : (v > 0.f ? v : 0.f)); }
    px[x] = (float)s; tx += (double)px[x];
  }
  for (int y = 0; y < H; ++y) {
    double s = 0.0;
    for (int x = 0; x < W; ++x) { float v = A[(size_t)y * W + x]; s += (double)(v != v ? v : (v > 0.f ? v : 0.f)); }
    py[y] = (float)s; ty += (double)py[y];
  }
  float dx = (float)tx; if (!(dx > 1e-6f)) dx = (dx != dx) ? dx : 1e-6f;
  float dy = (float)ty; if (!(dy > 1e-6f)) dy = (dy != dy) ? dy : 1e-6f;
  for (int x = 0; x < W; ++x) px[x] = px[x] / dx;
  for (int y = 0; y < H; ++y) py[y] = py[y] / dy;
}

/* ---- A8: x_hat = A^T (inv y); inv = (A A^T + eps I)^-1 given as doubles [Lo,Lo] ------------ */
API void oracle_right_inverse(const float* y, int Lo, int L, const double* inv, int clamp0, float* out) {
  float tmp[64];
  for (int k = 0; k < Lo; ++k) {
    double acc = 0.0;
    for (int j = 0; j < Lo; ++j) acc = acc + (double)y[j] * inv[(size_t)k * Lo + j];
    tmp[k] = (float)acc;
  }
  for (int l = 0; l < L; ++l) out[l] = 0.0f;
  for (int k = 0; k < Lo; ++k) {
    const int s = (int)(((long long)k * L) / Lo);
    const int e = (int)((((long long)(k + 1)) * L + Lo - 1) / Lo);
    const float a = 1.0f / (float)((e - s) > 1 ? (e - s) : 1);
    for (int l = s; l < e; ++l) out[l] = out[l] + tmp[k] * a;
  }
  if (clamp0) for (int l = 0; l < L; ++l) out[l] = out[l] > 0.0f ? out[l] : (out[l] != out[l] ? out[l] : 0.0f);
}

/* ---- A9 ------------------------------------------------------------------------------------ */
API void oracle_cdf_from_density(const float* p, int L, float* F) {
  double s = 0.0;
  for (int k = 0; k < L; ++k) {
    float v = p[k];
    v = (v != v || isinf(v)) ? 0.0f : (v > 0.0f ? v : 0.0f);
    F[k] = v; s += (double)v;
  }
  float den = (float)s; if (den < 1e-6f) den = 1e-6f;
  double c = 0.0;
  for (int k = 0; k < L; ++k) { c += (double)(F[k] / den); F[k] = (float)c; }
  F[L - 1] = 1.0f;
}

/* ---- np.interp(x = 0..n_out-1, xp, fp = 0..len-1) ------------------------------------------- */
static int search_with_guess(double key, const double* arr, int len, int guess) {
  int imin = 0, imax = len;
  if (key > arr[len - 1]) return len;
  if (key < arr[0]) return -1;
  if (len <= 4) { int i; for (i = 1; i < len && key >= arr[i]; ++i) {} return i - 1; }
  if (guess > len - 3) guess = len - 3;
  if (guess < 1) guess = 1;
  if (key < arr[guess]) {
    if (key < arr[guess - 1]) { imax = guess - 1; if (guess > 8 && key >= arr[guess - 8]) imin = guess - 8; }
    else return guess - 1;
  } else {
    if (key < arr[guess + 1]) return guess;
    if (key < arr[guess + 2]) return guess + 1;
    imin = guess + 2;
    if (guess < len - 8 - 1 && key < arr[guess + 8]) imax = guess + 8;
  }
  while (imin < imax) { const int imid = imin + ((imax - imin) >> 1); if (key >= arr[imid]) imin = imid + 1; else imax = imid; }
  return imin - 1;
}

static void interp_identity_fp(const double* xp, int len, int n_out, float* map) {
  int j = 0;
  for (int i = 0; i < n_out; ++i) {
    const double x = (double)i;
    double r;
    j = search_with_guess(x, xp, len, j);
    if (j == -1) r = 0.0;
    else if (j == len) r = (double)(len - 1);
    else if (j == len - 1) r = (double)j;
    else if (xp[j] == x) r = (double)j;
    else {
      const double slope = 1.0 / (xp[j + 1] - xp[j]);
      r = slope * (x - xp[j]) + (double)j;
      if (r != r) r = slope * (x - xp[j + 1]) + (double)(j + 1);
    }
    map[i] = (float)r;
  }
}

/* ---- A11 ----------------------------------------------------------------------------------- */
API void oracle_axis_map_from_cdf(const float* F, int L, int n_out, float* map) {
  const int len = L + 1;
  double* xn = (double*)malloc(sizeof(double) * len);
  xn[0] = 0.0;
  for (int k = 0; k < L; ++k) xn[k + 1] = (double)F[k] * (double)n_out;
  xn[len - 1] = (double)n_out;
  int tie = 0;
  for (int k = 0; k + 1 < len; ++k) if (xn[k + 1] - xn[k] <= 0.0) tie = 1;
  if (tie) {
    const float c = (float)(1e-4 / (double)(n_out > 1 ? n_out : 1));
    for (int k = 0; k < len; ++k) xn[k] += (double)(c * (float)k);
  }
  interp_identity_fp(xn, len, n_out, map);
  free(xn);
}

/* ---- A12: bilinear, replicate border, separable maps.  layout 0 = HWC, 1 = CHW ------------------ */
static inline float lerp_rn(float a, float b, float t) { float d = b - a; float m = t * d; return a + m; }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* mode 0 = exact (vertical lerp, then horizontal, unquantised coordinates);
 * mode 1 = cv2 (OpenCV's published remap algorithm: coordinates rounded to 1/32 pixel with cvRound, float32
 * sources use the 4 table weights (1-ty|ty)*(1-tx|tx), summed ((p00*w00 + p01*w01) + p10*w10) + p11*w11). */
static void axis_taps(const float* m, int n, int size, int mode, int* i0, int* i1, float* f) {
  for (int x = 0; x < n; ++x) {
    int i;
    if (mode == 1) {
      const float s = m[x] * 32.0f;
      /* cvRound on x86 (cvtss2si): round half to even; NaN or a value outside int32 gives INT_MIN */
      const long q = (s >= -2147483648.0f && s < 2147483648.0f) ? lrintf(s) : -2147483648L;
      i = (int)(q >> 5);
      f[x] = (float)(q & 31) * 0.03125f;
    } else {
      /* the coordinate is clamped to [-1, size] first (NaN counts as -1) */
      float mc = (m[x] != m[x]) ? -1.0f : m[x];
      mc = mc < -1.0f ? -1.0f : (mc > (float)size ? (float)size : mc);
      const float fl = floorf(mc);
      f[x] = mc - fl;
      i = (int)fl;
    }
    i0[x] = clampi(i, 0, size - 1); i1[x] = clampi(i + 1, 0, size - 1);
  }
}
static inline float cv2_sum(float p00, float p01, float p10, float p11, float fx, float fy) {
  const float ox = 1.0f - fx, oy = 1.0f - fy;
  const float w00 = oy * ox, w01 = oy * fx, w10 = fy * ox, w11 = fy * fx;
  const float a = p00 * w00, b = p01 * w01, c = p10 * w10, d = p11 * w11;
  float s = a + b; s = s + c; s = s + d;
  return s;
}

API void oracle_remap_bilinear_f32(const float* src, float* dst, int layout, int C, int H, int W, int Ho, int Wo,
                                   const float* mx, const float* my, int mode) {
  int* x0 = (int*)malloc(sizeof(int) * ((size_t)Wo * 2 + (size_t)Ho * 2));
  int* x1 = x0 + Wo; int* y0s = x1 + Wo; int* y1s = y0s + Ho;
  float* fx = (float*)malloc(sizeof(float) * ((size_t)Wo + Ho));
  float* fys = fx + Wo;
  axis_taps(mx, Wo, W, mode, x0, x1, fx);
  axis_taps(my, Ho, H, mode, y0s, y1s, fys);
  for (int y = 0; y < Ho; ++y) {
    const float fy = fys[y];
    const int y0 = y0s[y], y1 = y1s[y];
    for (int c = 0; c < (layout == 0 ? 1 : C); ++c) {
      const size_t cs = layout == 0 ? (size_t)C : 1;          /* element stride between neighbouring pixels */
      const int nc = layout == 0 ? C : 1;                     /* interleaved channels handled per pixel */
      const float* r0 = layout == 0 ? src + (size_t)y0 * W * C : src + ((size_t)c * H + y0) * W;
      const float* r1 = layout == 0 ? src + (size_t)y1 * W * C : src + ((size_t)c * H + y1) * W;
      float* o = layout == 0 ? dst + (size_t)y * Wo * C : dst + ((size_t)c * Ho + y) * Wo;
      for (int x = 0; x < Wo; ++x)
        for (int k = 0; k < nc; ++k) {
          const float p00 = r0[x0[x] * cs + k], p01 = r0[x1[x] * cs + k];
          const float p10 = r1[x0[x] * cs + k], p11 = r1[x1[x] * cs + k];
          if (mode == 1) {
            o[x * cs + k] = cv2_sum(p00, p01, p10, p11, fx[x], fy);
          } else {
            const float v0 = lerp_rn(p00, p10, fy);
            const float v1 = lerp_rn(p01, p11, fy);
            o[x * cs + k] = lerp_rn(v0, v1, fx[x]);
          }
        }
    }
  }
  free(x0);
  free(fx);
}

/* ---- uint8 sources, interleaved [H,W,C] (what cv2.remap receives in AGW/new_method.py:268-271) -------------------
 * mode 1 (cv2): int16 weights scaled by 2^15 -- round(w * 32768) saturated to int16, exact integers -- and
 * (sum + 2^14) >> 15; mode 0 (exact): the float32 lerps of the float path, rounded half to even, saturated. */
API void oracle_remap_bilinear_u8(const uint8_t* src, uint8_t* dst, int C, int H, int W, int Ho, int Wo, const float* mx,
                                  const float* my, int mode) {
  int* x0 = (int*)malloc(sizeof(int) * ((size_t)Wo * 2 + (size_t)Ho * 2));
  int* x1 = x0 + Wo; int* y0s = x1 + Wo; int* y1s = y0s + Ho;
  float* fx = (float*)malloc(sizeof(float) * ((size_t)Wo + Ho));
  float* fys = fx + Wo;
  axis_taps(mx, Wo, W, mode, x0, x1, fx);
  axis_taps(my, Ho, H, mode, y0s, y1s, fys);
  for (int y = 0; y < Ho; ++y) {
    const float fy = fys[y];
    const uint8_t* r0 = src + (size_t)y0s[y] * W * C;
    const uint8_t* r1 = src + (size_t)y1s[y] * W * C;
    uint8_t* o = dst + (size_t)y * Wo * C;
    for (int x = 0; x < Wo; ++x) {
      long w00 = 0, w01 = 0, w10 = 0, w11 = 0;
      if (mode == 1) {
        const float ox = 1.0f - fx[x], oy = 1.0f - fy;
        w00 = lrintf(oy * ox * 32768.0f); w01 = lrintf(oy * fx[x] * 32768.0f);
        w10 = lrintf(fy * ox * 32768.0f); w11 = lrintf(fy * fx[x] * 32768.0f);
        if (w00 > 32767) w00 = 32767;                        /* saturate_cast<short> of 1.0 * 2^15 */
      }
      for (int k = 0; k < C; ++k) {
        const int p00 = r0[(size_t)x0[x] * C + k], p01 = r0[(size_t)x1[x] * C + k];
        const int p10 = r1[(size_t)x0[x] * C + k], p11 = r1[(size_t)x1[x] * C + k];
        long v;
        if (mode == 1) {
          v = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1L << 14)) >> 15;
        } else {
          const float v0 = lerp_rn((float)p00, (float)p10, fy);
          const float v1 = lerp_rn((float)p01, (float)p11, fy);
          v = lrintf(lerp_rn(v0, v1, fx[x]));
        }
        o[(size_t)x * C + k] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
      }
    }
  }
  free(x0);
  free(fx);
}

/* ---- whole hot path for one image (float32): attention stack -> warped image ------------------- */
API void oracle_warp_from_attention_stack_f32(const float* img, float* out, int layout, int C, int H, int W,
                                              const float* rows /* [T,1,heads,kv] */, int T, int heads, int kv,
                                              int start, const double* inv_x, const double* inv_y, int mode) {
  float att[576], px[24], py[24];
  const int32_t st = start;
  oracle_attn_reduce_stack_f32(rows, T, 1, heads, kv, &st, 576, att);
  oracle_marginals(att, 24, 24, px, py);
  float* dx = (float*)malloc(sizeof(float) * (size_t)(2 * W + 2 * H));
  float* dy = dx + W; float* mx = dy + H; float* my = mx + W;
  oracle_right_inverse(px, 24, W, inv_x, 1, dx);
  oracle_right_inverse(py, 24, H, inv_y, 1, dy);
  oracle_cdf_from_density(dx, W, dx);
  oracle_cdf_from_density(dy, H, dy);
  oracle_axis_map_from_cdf(dx, W, W, mx);
  oracle_axis_map_from_cdf(dy, H, H, my);
  oracle_remap_bilinear_f32(img, out, layout, C, H, W, H, W, mx, my, mode);
  free(dx);
}
