/*
 * attwarp.h -- C ABI of libattwarp_hip.so, the MI355X (gfx950) implementation of
 * AttWarp's attention-guided image-warping hot path.
 *
 * The reference (dwipddalal/AttWarp) is pure Python: it has no FFI, custom-op or
 * plugin layer for this path, so there is no existing binding to mirror.  Each
 * entry point below replaces one reference *function* (cited as file:line,
 * AGW = "Attention Guided Warping", MN = "model/marginalnet_full_dataset"); the
 * Python package attwarp_amd re-exposes the reference's names on top of these
 * through ctypes (see INTEGRATION.md for the stub a maintainer would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter comment says "host";
 *   - all tensors are dense, row-major ("C contiguous") with the shape given in
 *     the comment, except attwarp_attn_reduce_step which takes element strides;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); every
 *     call only enqueues work on that stream, never synchronises, never
 *     allocates: outputs and workspaces are owned by the caller;
 *   - return value: 0 on success, a negative ATTWARP_E_* code on failure (nothing
 *     is enqueued then); attwarp_last_error() gives a thread-local message;
 *   - the library keeps no global mutable state on the product path: transform selection etc. are
 *     arguments (the reference keeps them in module globals, AGW/new_method.py:159-163,378-403) and no
 *     environment variable is read; the only process-wide setting is the test hook attwarp_debug_set().
 */
#ifndef ATTWARP_H
#define ATTWARP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ATTWARP_VERSION 100 /* 0.1.0 */

#if defined(__GNUC__)
#define ATTWARP_API __attribute__((visibility("default")))
#else
#define ATTWARP_API
#endif

/* element types */
enum { ATTWARP_F32 = 0, ATTWARP_F16 = 1, ATTWARP_BF16 = 2, ATTWARP_U8 = 3, ATTWARP_F64 = 4 };
/* image layouts: HWC = [B,H,W,C] interleaved (OpenCV / numpy), CHW = [B,C,H,W] planar (torch) */
enum { ATTWARP_HWC = 0, ATTWARP_CHW = 1 };
/* resample arithmetic: CV2 = what cv2.remap(INTER_LINEAR) computes per OpenCV's published algorithm: coordinates
 * rounded to 1/32 pixel, 4 table weights for float32, 15-bit fixed point for uint8 (the reference's arithmetic,
 * the default of the numpy / uint8 drop-ins; unpinned); EXACT = bilinear on the unquantised
 * coordinates (= F.grid_sample(bilinear, border, align_corners=True), north_star's named op).
 * Both modes run on the same staged kernels. */
enum { ATTWARP_EXACT = 0, ATTWARP_CV2 = 1 };
/* attention transforms, AGW/new_method.py:134-179 */
enum { ATTWARP_T_IDENTITY = 0, ATTWARP_T_SQUARE = 1, ATTWARP_T_SQRT = 2, ATTWARP_T_EXP = 3, ATTWARP_T_LOG = 4 };

/* error codes */
enum {
  ATTWARP_OK = 0,
  ATTWARP_E_ARG = -1,      /* bad argument (null pointer, non-positive size, unknown enum) */
  ATTWARP_E_UNSUPPORTED = -2, /* valid request this build cannot serve (size limit, dtype) */
  ATTWARP_E_LAUNCH = -3    /* HIP reported an error while enqueueing */
};

ATTWARP_API int attwarp_version(void);
ATTWARP_API const char* attwarp_last_error(void);

/* ---- test / measurement hook: exists ONLY in the tuning flavour of the library (libattwarp_hip_tuning.so, built with
 * -DATTWARP_TUNING; the product library libattwarp_hip.so does not export it and contains no override table).
 * Kernel variants are chosen automatically from the shapes; the parity tests and the A/B tools force a particular one
 * to check the variants against each other (e.g. key "remap_variant" = 1: generic gather kernel only; "remap_rows" =
 * R).  value < 0 restores "automatic", key "reset" restores every key; the previous value of the key is returned
 * through *previous when it is not NULL.  Keys: see kTuneNames in csrc/error.hip.  Process wide, single threaded,
 * test only. */
#ifdef ATTWARP_TUNING
ATTWARP_API int attwarp_debug_set(const char* key, int value, int* previous);
/* measurement yardstick (csrc/calib.hip): ONE launch that reads read_bytes from src and writes write_bytes to dst (16-byte
 * aligned device buffers) -- a plain streaming kernel for the bytes of a chain step, the ruler of bench.py's `step_over_copy` */
ATTWARP_API int attwarp_debug_stream_copy(const void* src, size_t read_bytes, void* dst, size_t write_bytes, void* stream);
#endif

/* ---- A1: BatchMaskHookLogger._process_attention, AGW/attention_extraction/llava.py:385-396
 * attn [B,heads,q,kv] with element strides; for sample b uses row q-1, columns
 * starts[b] .. starts[b]+ntok-1; per-head renormalisation (x / (sum + 1e-12)), mean over heads.
 * out [B,ntok] in the same dtype (F32/F16/BF16).  starts: device int32[B]; ntok <= kv_len is required and a start
 * outside [0, kv_len-ntok] is clamped into it on the device (no out-of-row reads).
 * F16 / BF16 rows with unit kv stride whose slice starts at an odd element are read in dword-aligned four-element words:
 * the element before the slice and up to three elements behind it are read with it -- elements of the same row (a slice
 * that ends within three elements of the row's end is read with plain element-aligned loads instead), so nothing outside
 * the allocation the tensor lives in is touched (the element before the slice of a view that starts at an odd element of
 * its buffer is the only access outside the tensor itself). */
ATTWARP_API int attwarp_attn_reduce_step(const void* attn, int dtype, int B, int heads, int q_len, int kv_len,
                             int64_t stride_b, int64_t stride_h, int64_t stride_q, int64_t stride_kv,
                             const int32_t* starts, int ntok, void* out, void* stream);

/* ---- A2: BatchMaskHookLogger.finalize_batch, llava.py:401-411
 * steps [T,B,ntok] -> out [B,ntok] = mean over T (same dtype). */
ATTWARP_API int attwarp_attn_finalize(const void* steps, int dtype, int T, int B, int ntok, void* out, void* stream);

/* ---- A1+A2 fused over a captured stack of last-query rows.
 * rows [T,B,heads,kv] -> out [B,ntok]; equals finalize(step(rows[t]) for t).
 * ws: workspace of attwarp_attn_reduce_stack_workspace_bytes(...) bytes ([T,B,ntok] step maps). */
ATTWARP_API size_t attwarp_attn_reduce_stack_workspace_bytes(int dtype, int T, int B, int ntok);
ATTWARP_API int attwarp_attn_reduce_stack(const void* rows, int dtype, int T, int B, int heads, int kv_len,
                              const int32_t* starts, int ntok, void* out, void* ws, void* stream);

/* ---- "next" row 4: hook-side capture without materialising [B,heads,q,kv] probabilities.
 * Replaces  register_hook_and_patch (llava.py:422-438: output_attentions=True on the target layer, i.e. HF's
 * eager_attention_forward producing softmax(QK^T*scaling + mask) for every query row) followed by
 * _process_attention (llava.py:385-396), which only consumes the LAST query row.
 *   q  : post-RoPE query vectors of the last token, [B,heads,head_dim], element strides q_stride_b/_h
 *   k  : post-RoPE key cache, [B,kv_heads,kv_len,head_dim], element strides k_stride_b/_h/_t
 *        (head_dim contiguous; heads % kv_heads == 0: grouped-query attention shares a key head)
 *   kv_begin : device int32[B] or NULL - first attended key position (left padding); earlier keys get p = 0
 *   starts   : device int32[B], image-token slice starts, clamped into [0, kv_len-ntok] on the device
 *   out      : [B,ntok] in `dtype` = mean over heads of p[st:st+ntok] / (sum + 1e-12), as A1
 *   ws       : attwarp_attn_probe_workspace_bytes(...) bytes, 16-byte aligned
 * Rounding mirrors the eager path's dtype transitions (matmul -> dtype, *scaling -> dtype, softmax in
 * float32 -> dtype).  q, k 16-byte aligned, strides multiples of 16 bytes, kv_len <= 15872, ntok <= 1024. */
ATTWARP_API size_t attwarp_attn_probe_workspace_bytes(int dtype, int B, int heads, int ntok);
ATTWARP_API int attwarp_attn_probe_last_query(const void* q, const void* k, int dtype, int B, int heads, int kv_heads,
                                  int head_dim, int kv_len, int64_t q_stride_b, int64_t q_stride_h,
                                  int64_t k_stride_b, int64_t k_stride_h, int64_t k_stride_t,
                                  const int32_t* kv_begin, const int32_t* starts, int ntok, float scaling,
                                  void* out, void* ws, void* stream);

/* ---- A3: revise_mask = normalize("min") -> enhance -> k x k box filter, llava.py:207-238
 * mask [B,n,n] float32 -> out [B,n,n] float32 (n <= 32, odd kernel_size <= 7). */
ATTWARP_API int attwarp_mask_postproc(const float* mask, int B, int n, int kernel_size, float enhance_coe,
                          float* out, void* stream);

/* ---- A4: ToPILImage (x*255, truncating cast) + PIL Image.resize(LANCZOS), llava.py:192-196,243,253
 * mask_f32 [B,h,w] float32 in [0,1]  (or mask_u8 [B,h,w] if mask_f32 is NULL)
 * -> out [B,out_h,out_w] uint8.  Coefficients: host-computed Pillow tables uploaded by the
 * caller: bounds_* int32[out,2] = (first tap, tap count), kk_* int32[out,ksize] (22-bit fixed point; entries at
 * index >= tap count are never used: every kernel form masks them by the row's tap count; a row stride of exactly 8
 * -- pad narrower tables to 8 columns with anything -- selects the fastest vertical pass).
 * tmp: uint8 [B,h,out_w] workspace (horizontal pass output). */
ATTWARP_API int attwarp_mask_upsample_lanczos(const float* mask_f32, const uint8_t* mask_u8, int B, int h, int w,
                                  int out_h, int out_w,
                                  const int32_t* bounds_x, const int32_t* kk_x, int ksize_x,
                                  const int32_t* bounds_y, const int32_t* kk_y, int ksize_y,
                                  uint8_t* tmp, uint8_t* out, void* stream);

/* ---- "next" row 3: warped uint8 image -> CLIP-ready tensor, replacing the PNG round trip
 * AGW/new_method.py:491 (cv2.imwrite) -> AGW/evaluate_accuracy.py:157-158 (Image.open, process_images):
 * HF CLIPImageProcessor as LLaVA-1.5 configures it = PIL BICUBIC resize (shorter edge -> size), center crop,
 * float32(float64(u8)*(1/255)), (x-mean)/std in float32, channels first.
 * src [B,h,w,C] uint8 (C<=4) -> out [B,C,size,size] (F32 or F16).  The caller passes Pillow's 8-bit coefficient
 * tables for the resized width / height (identity tables when an axis keeps its size) and the crop origin
 * (top,left) in the resized image; mean/stdv are HOST arrays of C floats; tmp: uint8 [B,h,size,C]. */
ATTWARP_API int attwarp_clip_preprocess_u8(const uint8_t* src, int B, int h, int w, int C, int top, int left, int size,
                               const int32_t* bounds_x, const int32_t* kk_x, int ksize_x,
                               const int32_t* bounds_y, const int32_t* kk_y, int ksize_y,
                               const float* mean, const float* stdv, uint8_t* tmp, void* out, int out_dtype,
                               void* stream);

/* ---- "next" row 1: the memory-bound tail of MarginalNet.forward, MN/model.py:73-88.
 * masked_token_mean: tok [B,Lt,D] (F32/F16/BF16), mask [B,Lt] float32 (the reference's [B,Lt,1])
 *   -> out [B,D] float32 = sum_l(float(tok)*mask) / max(sum_l mask, 1)            (model.py:77-78)
 * film_axis_means: v [B,Ch,H,W] float32, gamma_beta [B,2*Ch] float32 (film output: gamma | beta)
 *   -> vx [B,Ch,W] = mean over H of (gamma*v + beta), vy [B,Ch,H] = mean over W   (model.py:80-88)
 *   H*(W+1) <= 4096. */
ATTWARP_API int attwarp_masked_token_mean(const void* tok, int dtype, const float* mask, int B, int Lt, int D,
                              float* out, void* stream);
ATTWARP_API int attwarp_film_axis_means(const float* v, const float* gamma_beta, int B, int Ch, int H, int W,
                            float* vx, float* vy, void* stream);

/* ---- A5: F.adaptive_avg_pool2d(A,(oh,ow)), call sites MN/trainer.py:197,433,465
 * A [B,H,W] float32 -> out [B,oh,ow] float32.  sanitize != 0 also applies the trainer's
 * torch.nan_to_num(A, nan=0, posinf=0, neginf=0).clamp_min(0) (MN/trainer.py:202) to the pooled map. */
ATTWARP_API int attwarp_adaptive_avg_pool(const float* A, int B, int H, int W, int oh, int ow, int sanitize, float* out,
                              void* stream);

/* ---- A6: gt_marginals, MN/checkpoint_utils.py:43-51
 * A [B,H,W] float32 -> px [B,W], py [B,H] float32.  ws: workspace of
 * attwarp_axis_sums_workspace_bytes(B,H,W) bytes. */
ATTWARP_API size_t attwarp_axis_sums_workspace_bytes(int B, int H, int W);
ATTWARP_API int attwarp_gt_marginals(const float* A, int B, int H, int W, float* px, float* py, void* ws, void* stream);

/* ---- A7: safe_softmax(dim=1), MN/model.py:8-14.  logits [B,N] -> out [B,N], N <= 4096. */
ATTWARP_API int attwarp_safe_softmax(const float* logits, int B, int N, float eps, float* out, void* stream);

/* ---- A8: upsample_pdf_right_inverse, MN/checkpoint_utils.py:64-131
 * y [N,Lo] float32 -> out [N,L] float32.  inv: device float64 [Lo,Lo] = (A A^T + eps I)^-1 of the
 * adaptive-avg-pool1d matrix A[Lo,L] (24x24: computed once per (Lo,L,eps) on the host). Lo <= 64. */
ATTWARP_API int attwarp_upsample_pdf_right_inverse(const float* y, int N, int Lo, int L, const double* inv,
                                       float* out, void* stream);

/* ---- A9: cdf_from_density, MN/checkpoint_utils.py:30-41.  p [B,L] -> F [B,L] float32. */
ATTWARP_API int attwarp_cdf_from_density(const float* p, int B, int L, float* F, void* stream);

/* ---- A10: _make_strictly_increasing (:17-28) and resample_cdf (:53-62). */
ATTWARP_API int attwarp_make_strictly_increasing(const float* F, int B, int N, double eps, float* out, void* stream);
ATTWARP_API int attwarp_resample_cdf(const float* F, int B, int N, int L, float* out, void* stream);

/* ---- A11: grid construction of warp_from_cdf_torch, MN/checkpoint_utils.py:167-193
 * F [B,L] float32 CDF -> map [B,n_out] float32 source coordinate per output index
 * (the reference meshgrids two such vectors into dense maps for cv2.remap). L <= 16384. */
ATTWARP_API int attwarp_axis_map_from_cdf(const float* F, int B, int L, int n_out, float* map, void* stream);

/* ---- A8+A9+A11 fused: 24-bin PDFs -> maps, the MarginalNet inference chain MN/trainer.py:285-289
 * px [B,Lo], py [B,Lo] -> map_x [B,W_out], map_y [B,H_out]; inv_x / inv_y as in A8 for L=W / L=H. */
ATTWARP_API int attwarp_axis_maps_from_pdf(const float* px, const float* py, int B, int Lo, int W, int H,
                               int W_out, int H_out, const double* inv_x, const double* inv_y,
                               float* map_x, float* map_y, void* stream);

/* ---- A2+A6+A8+A9+A11 fused: per-step attention maps (A1 output) -> inverse maps in ONE launch.
 * steps [T,B,g*g] float32 -> mean over steps -> marginals of the g x g map -> PDF up-sample -> CDF ->
 * map_x [B,W_out], map_y [B,H_out]; att_out (optional, may be NULL): the aggregated [B,g*g] map (A2 output).
 * Bit-identical to the separate stages. */
ATTWARP_API int attwarp_axis_maps_from_steps(const float* steps, int T, int B, int g, int W, int H, int W_out, int H_out,
                                 const double* inv_x, const double* inv_y, float* map_x, float* map_y,
                                 float* att_out, void* stream);

/* the same for step maps in the model dtype (F32 / F16 / BF16): the mean over steps (A2) is accumulated in double,
 * rounded ONCE to that dtype and divided by T in it, as attwarp_attn_finalize does; the rest is float32.  Equals
 * attwarp_attn_finalize -> float() -> attwarp_gt_marginals -> attwarp_axis_maps_from_pdf bit for bit. */
ATTWARP_API int attwarp_axis_maps_from_steps_t(const void* steps, int dtype, int T, int B, int g, int W, int H, int W_out,
                                   int H_out, const double* inv_x, const double* inv_y, float* map_x, float* map_y,
                                   float* att_out, void* stream);

/* ---- one step of a STREAM of equally shaped batches in ONE launch (pipeline.OverlappedWarp; replaces the three
 * launches reduce -> maps -> resample of AGW/main_batched.py's per-batch chain when batches follow each other):
 *   R  the A12 resample of batch k      src [B,...] float32 + map_x/map_y (built by the previous step)  -> dst
 *   M  the maps of batch k+1            steps_in [T,B,g*g] (previous step's reduce) -> map_x_next / map_y_next
 *   A  the A1 reduce of batch k+2       rows [n_rows = T*B, heads, kv_len], row j uses starts[j % starts_mod]
 *                                       -> steps_out [T*B, ntok]
 * rows, steps_in and steps_out share attn_dtype (F32 / F16 / BF16: the model dtype, as the hook delivers them; A2's
 * mean over steps is rounded in that dtype like llava.py:409-411);
 * as block ranges of one grid (map blocks first, then reduce and resample blocks interleaved in chunks of 8): no
 * launch boundary, no queue hand-off inside a step.  The three pieces work on different batches and must not alias
 * (steps_out != steps_in, map_*_next != map_*).  M (steps_in == NULL) and A (rows == NULL) are optional.  Same
 * arithmetic, bit for bit, as attwarp_remap_bilinear / attwarp_axis_maps_from_steps / attwarp_attn_reduce_step.
 * ATTWARP_E_UNSUPPORTED when the image shape takes the generic resample (rows wider than 4096 floats, unaligned
 * rows), ntok is not a multiple of 4 or > 768, g > 32, ntok != g*g while both steps_in and rows are given (steps_out of
 * one call is steps_in [T,B,g*g] of the next), or an axis so long that the map construction needs more than 64 KB of
 * LDS (max(W,H) > ~4800): use the three separate entry points then. */
ATTWARP_API int attwarp_warp_step_fused(const float* src, float* dst, int layout, int B, int C, int H, int W, int H_out,
                            int W_out, const float* map_x, const float* map_y, int mode,
                            int attn_dtype, const void* steps_in, int T, int g, const double* inv_x,
                            const double* inv_y, float* map_x_next, float* map_y_next,
                            const void* rows, int n_rows, int heads, int kv_len, const int32_t* starts,
                            int starts_mod, int ntok, void* steps_out, void* stream);

/* ---- the same one-launch step for TWO consecutive batches of the stream per piece ("slots"): R(k), R(k+1) | M(k+2),
 * M(k+3) | A(k+4), A(k+5) -- every dependency is on an earlier launch, and the launch's ramp and tail are paid once per two
 * batches.  `slots` is a HOST array of nslots (1 or 2) pointer sets (device pointers, meanings as the arguments of
 * attwarp_warp_step_fused; `starts` per slot); geometry, dtype and mode are shared.  nslots == 1 equals
 * attwarp_warp_step_fused.  Both slots must carry the same pieces, write different buffers, and their images must share
 * their 16-byte alignment. */
typedef struct attwarp_step_slot {
  const float* src; float* dst; const float* map_x; const float* map_y;      /* R: resample of this slot's batch       */
  const void* steps_in; float* map_x_next; float* map_y_next;                 /* M: step maps [T,B,g*g] -> its maps     */
  const void* rows; const int32_t* starts; void* steps_out;                   /* A: rows [n_rows,heads,kv] -> step maps */
} attwarp_step_slot;
ATTWARP_API int attwarp_warp_step_fused_slots(const attwarp_step_slot* slots, int nslots, int layout, int B, int C, int H, int W,
                                  int H_out, int W_out, int mode, int attn_dtype, int T, int g, const double* inv_x,
                                  const double* inv_y, int n_rows, int heads, int kv_len, int starts_mod, int ntok,
                                  void* stream);

/* ---- the attention reduce of batch k+2 and the map construction of batch k+1 of a batch stream in ONE launch (for
 * large images, where the resample keeps its own launch): rows [n_rows = T*B, heads, kv_len] -> steps_out [T*B, ntok];
 * steps_in [T,B,g*g] (the previous call's steps_out buffer of the other parity) -> map_x [B,W_out], map_y [B,H_out].
 * One dtype (F32 / F16 / BF16) for rows and both step buffers; ntok == g*g.  Equals attwarp_attn_reduce_step +
 * attwarp_axis_maps_from_steps_t bit for bit. */
ATTWARP_API int attwarp_attn_reduce_and_maps(int attn_dtype, const void* rows, int n_rows, int heads, int kv_len,
                                 const int32_t* starts, int starts_mod, int ntok, void* steps_out,
                                 const void* steps_in, int T, int B, int g, int W, int H, int W_out, int H_out,
                                 const double* inv_x, const double* inv_y, float* map_x, float* map_y, void* stream);

/* ---- one step of the main_batched chain (AGW/main_batched.py:243-287: revise_mask -> x255 uint8 -> PIL LANCZOS ->
 * float64 marginals -> CDF -> np.interp -> uint8 cv2.remap) for a STREAM of equally shaped batches, as ONE launch, with
 * save_warped_image's own keyword arguments (AGW/new_method.py:405-411: transform, exp_scale, exp_divisor, apply_inverse;
 * :134-191 the five transforms and their inverses, :219-226 the inverse on the marginals; main_batched.py:280-287 passes
 * "identity", 1.0, 1.0, False; the module default is "sqrt", :191).  The five stages run on five different batches (block
 * ranges of one grid):
 *   R(k)    images [B,H,W,C] uint8 + map_x [B,W_out], map_y [B,H_out]         -> out [B,H_out,W_out,C]   (mode cv2)
 *   F(k+1)  sums_in (axis sums of the mask of batch k+1, written by P)        -> map_x_next, map_y_next
 *   P(k+2)  mota_in [B,H,W] uint8 (up-sampled mask of batch k+2)              -> sums_out
 *   L(k+3)  rev_in [B,g,g] float32 (revised mask of batch k+3) + Pillow tables (ksize_y == 8, zero padded) -> mota_out
 *   V(k+4)  masks [B,g,g] float32 (aggregated attention of batch k+4)         -> rev_out
 * sums_*: attwarp_axis_sums_workspace_bytes(B,H,W) bytes each.  An output buffer must not alias the buffer the next
 * stage reads in the same launch (double buffer each intermediate by batch parity).  transform = ATTWARP_T_* applies in
 * P (element transform) and F (apply_inverse: the inverse on the marginals); transform_lut: for ATTWARP_T_SQRT / EXP / LOG
 * the 256 doubles attwarp_attention_transform_lut wrote for the same (transform, exp_scale, exp_divisor) -- a constant of the
 * stream, computed once -- NULL allowed for identity / square.  Every stage computes what
 * attwarp_mask_postproc / attwarp_mask_upsample_lanczos / attwarp_axis_maps_from_attention(U8, transform, ...) /
 * attwarp_remap_bilinear(U8, HWC, CV2) compute, bit for bit.  ATTWARP_E_UNSUPPORTED when one of the stages would not
 * run on its staged kernel for this shape (rows wider than 4096 bytes, W not a multiple of 4, W or H equal to g, ...):
 * use the separate entry points, or attwarp_mask_chain_ragged (below: any width), then. */
ATTWARP_API int attwarp_mask_chain_step(const uint8_t* images, uint8_t* out, int B, int C, int H, int W, int H_out, int W_out,
                            const float* map_x, const float* map_y,
                            const void* sums_in, float* map_x_next, float* map_y_next,
                            const uint8_t* mota_in, void* sums_out,
                            const float* rev_in, const int32_t* bounds_x, const int32_t* kk_x, int ksize_x,
                            const int32_t* bounds_y, const int32_t* kk_y, int ksize_y, uint8_t* mota_out,
                            const float* masks, int g, int kernel_size, float enhance_coe, float* rev_out,
                            int transform, double exp_scale, double exp_divisor, int apply_inverse,
                            const double* transform_lut, void* stream);

/* The 256-entry table of the transformed byte values of a uint8 attention map: lut[v] = transform(max(v, 0)) + 1e-9 for
 * v = 0 .. 255 (AGW/new_method.py:134-179,208-215), by the very device functions attwarp_axis_maps_from_attention(U8) uses
 * (bit-identical).  lut: 256 doubles on the device.  Needed by the one-launch chain steps for ATTWARP_T_SQRT / EXP / LOG. */
ATTWARP_API int attwarp_attention_transform_lut(int transform, double exp_scale, double exp_divisor, double* lut, void* stream);

/* ---- the same chain for batches of DIFFERENTLY sized images: what AGW/main_batched.py:243-287 actually holds (`b_images[j]`
 * are PIL images at their native sizes; blend_mask up-samples the mask to `image.size`, llava.py:253; save_warped_image warps
 * at that size to width x height = 500 x 500, new_method.py:415-422,478-488).  Output is dense: [B,H_out,W_out,C].
 *
 * A batch is described by a TABLE the host builds once per batch and copies to the device unchanged (it is position
 * independent):
 *   1. fill attwarp_ragged_image[B] (HOST array of DEVICE pointers): the image, its size, and Pillow's 8-bit LANCZOS
 *      coefficient tables for g -> W (ksize_x <= 8 columns) and g -> H (exactly 8 columns, zero padded) --
 *      attwarp_pil_coeffs_8bpc computes them on the host, callers cache them per distinct size;
 *   2. attwarp_ragged_table_bytes(...) -> size; attwarp_ragged_plan(...) writes the table into HOST memory: an
 *      attwarp_ragged_header (public, below: buffer sizes the caller must provide) followed by per-image records, numpy's
 *      pairwise-summation plans of the distinct widths / heights, the block maps of the stages whose block count
 *      depends on the image and the order in which the stages visit the images (largest first);
 *   3. copy table_bytes to the device (any 8-byte aligned address);
 *   4. attwarp_mask_chain_ragged(...) with the host AND the device copy of each stage's table.
 * Limits (ATTWARP_E_UNSUPPORTED from attwarp_ragged_plan otherwise; use the per-image entry points then): rows of 4 ..
 * 4096 bytes on both sides (W*C, W_out*C), H, W > g (the mask is up-sampled on both axes), max(H,W) <= 8192, g <= 32,
 * B <= 65535.  Any alignment: 683 x 3-byte rows are served by the same staged kernels as 684 x 3. */
#define ATTWARP_RAGGED_MAX_LEAVES 64
enum { ATTWARP_PIL_LANCZOS = 0, ATTWARP_PIL_BICUBIC = 1 };
typedef struct attwarp_ragged_image {
  const uint8_t* image;                            /* device: [H,W,C] uint8 interleaved, any byte address */
  const int32_t* bounds_x; const int32_t* kk_x;    /* device: Pillow tables g -> W: int32 [W,2], int32 [W,ksize_x] */
  const int32_t* bounds_y; const int32_t* kk_y;    /* device: g -> H: int32 [H,2], int32 [H,8] (zero padded to 8 columns) */
  int32_t H, W, ksize_x, reserved;
} attwarp_ragged_image;
typedef struct attwarp_ragged_header {             /* first bytes of a table written by attwarp_ragged_plan */
  uint32_t magic;
  int32_t B, C, g, H_out, W_out;
  int32_t nL, nP, nR, nplans;                      /* blocks of the up-sampling, marginals and resample stages; distinct plans */
  int32_t rows_per_block, blocks_per_image, kd, max_hw;
  uint64_t table_bytes;                            /* size of the table (copy this many bytes to the device) */
  uint64_t mota_bytes;                             /* the batch's up-sampled masks, packed: uint8 buffer of this size */
  uint64_t sums_bytes;                             /* the batch's axis-sum workspace (float64) */
  uint64_t lds_bytes;
  uint64_t off_images, off_plans, off_lmap, off_pmap, off_order;
} attwarp_ragged_header;

/* HOST: Pillow's ImagingResample coefficient tables for an 8-bit image (precompute_coeffs + normalize_coeffs_8bpc,
 * libImaging/Resample.c; 22-bit fixed point) for resizing in_size -> out_size with `filter`.  bounds: int32 [out_size,2]
 * = (first tap, tap count); kk: int32 [out_size,kk_cols], zero padded.  Returns the tap-count bound ksize (<= kk_cols),
 * 1 with identity tables when in_size == out_size (Pillow skips that pass), or a negative error code. */
ATTWARP_API int attwarp_pil_coeffs_8bpc(int in_size, int out_size, int filter, int32_t* bounds, int32_t* kk, int kk_cols);

/* HOST: size of / contents of the table of one batch (see above).  0 / a negative code when the batch does not run on
 * the ragged kernel (attwarp_last_error() says which image and why). */
ATTWARP_API size_t attwarp_ragged_table_bytes(const attwarp_ragged_image* images, int B, int C, int g, int H_out, int W_out);
ATTWARP_API int attwarp_ragged_plan(const attwarp_ragged_image* images, int B, int C, int g, int H_out, int W_out, void* table,
                        size_t table_bytes);

/* One launch running up to five stages, each on ITS OWN batch (as attwarp_mask_chain_step; a stage whose host table --
 * V: whose `masks` -- is NULL is skipped, so one batch alone is five calls with one stage each):
 *   R(k)    r_*: images of the table + map_x [B,W_out], map_y [B,H_out]          -> out [B,H_out,W_out,C]   (mode cv2)
 *   F(k+1)  f_*: sums_in (sums_bytes of that table, written by P)               -> map_x_next, map_y_next
 *   P(k+2)  p_*: mota_in (mota_bytes of that table, written by L)               -> sums_out
 *   L(k+3)  l_*: rev_in [B,g,g] float32 (written by V) + the images' Pillow tables -> mota_out
 *   V(k+4)  masks [B_masks,g,g] float32                                          -> rev_out
 * transform / exp_scale / exp_divisor / apply_inverse / transform_lut: as attwarp_mask_chain_step (used by the P and F stages).
 * Every stage computes what attwarp_mask_postproc / attwarp_mask_upsample_lanczos / attwarp_axis_maps_from_attention(U8,
 * transform, ...) / attwarp_remap_bilinear(U8, HWC, CV2) compute on that image alone, bit for bit.  The batches of one
 * launch share C, g and the output size; their B and image sizes are free. */
ATTWARP_API int attwarp_mask_chain_ragged(const void* r_host, const void* r_dev, uint8_t* out, const float* map_x, const float* map_y,
                              const void* f_host, const void* f_dev, const void* sums_in, float* map_x_next, float* map_y_next,
                              const void* p_host, const void* p_dev, const uint8_t* mota_in, void* sums_out,
                              const void* l_host, const void* l_dev, const float* rev_in, uint8_t* mota_out,
                              const float* masks, int B_masks, int g, int kernel_size, float enhance_coe, float* rev_out,
                              int transform, double exp_scale, double exp_divisor, int apply_inverse,
                              const double* transform_lut, void* stream);

/* ---- A13: grid construction of warp_image_by_attention, AGW/new_method.py:206-265
 * att [B,h,w] (U8/F32/F64) -> map_x [B,new_w], map_y [B,new_h] float32.
 * ws: workspace of attwarp_axis_sums_workspace_bytes(B,h,w) bytes. */
ATTWARP_API int attwarp_axis_maps_from_attention(const void* att, int dtype, int B, int h, int w, int new_w, int new_h,
                                     int transform, double exp_scale, double exp_divisor, int apply_inverse,
                                     float* map_x, float* map_y, void* ws, void* stream);

/* ---- cv2.resize(image, (W_out, H_out), interpolation=cv2.INTER_LINEAR), AGW/new_method.py:369 (reached from :478
 * when the attention map's size differs from the image's; a no-op in both reference drivers).
 * src [B,H,W,C] interleaved, F32 or U8 -> dst [B,H_out,W_out,C].  OpenCV's published algorithm (resize.cpp): half-pixel
 * centres, uint8 with 11-bit fixed-point coefficients and its (b * (D >> 4)) >> 16 vertical pass, an exact 2 x 2
 * decimation as INTER_AREA; parity unpinned like attwarp_remap_bilinear's cv2 mode (OpenCV is absent here). */
ATTWARP_API int attwarp_resize_linear(const void* src, void* dst, int dtype, int B, int C, int H, int W, int H_out,
                          int W_out, void* stream);

/* ---- A12 / A13 tail: cv2.remap(INTER_LINEAR, BORDER_REPLICATE) with separable maps,
 * AGW/new_method.py:268-271, MN/checkpoint_utils.py:195-198.
 * src [B,H,W,C] (HWC) or [B,C,H,W] (CHW), dtype F32, U8 or F64 -> dst same layout with (H_out,W_out).
 * map_x [B,W_out], map_y [B,H_out] float32 source coordinates.  C <= 4.  (F64: the dtype pass-through of
 * warp_from_cdf_torch, MN/checkpoint_utils.py:152,203 -- OpenCV's CV_64F arithmetic: float32 table weights, the four
 * products accumulated in double; generic gather kernel only.)
 * Coordinates outside the image take the replicate border.  Non-finite / huge coordinates: mode CV2 follows
 * cvRound(32 * m) of OpenCV's x86 builds (NaN, +-Inf and products outside int32 -> INT_MIN -> pixel 0, zero fraction,
 * for either sign); mode EXACT clamps the coordinate to [-1, size] first (NaN counts as -1).  No output is NaN
 * unless a source pixel is. */
ATTWARP_API int attwarp_remap_bilinear(const void* src, void* dst, int dtype, int layout, int B, int C, int H, int W,
                           int H_out, int W_out, const float* map_x, const float* map_y, int mode,
                           void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ATTWARP_H */
