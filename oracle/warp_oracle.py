"""CPU oracle for the AttWarp attention-guided warp hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product path (``attwarp_amd``) never routes through this module; it
fails loudly when the HIP library is missing.

It is a numpy restatement (no torch, no OpenCV) of the reference algorithm.
Every function cites the reference lines it follows (paths relative to the
upstream checkout; ``AGW`` = ``Attention Guided Warping``, ``MN`` =
``model/marginalnet_full_dataset``).

Pinning status
--------------
* Everything up to and including the 1-D inverse maps handed to ``cv2.remap``
  is pinned to the reference itself: ``tests/golden/make_golden.py`` imports
  the reference by path in the build container and stores its outputs under
  ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this module
  against them.
* The mask up-sample (PIL ``Image.resize(LANCZOS)``) is pinned to Pillow 12.2.0
  golden vectors (bit-exact).
* The final resample (``cv2.remap``) is **parity unpinned**: OpenCV is neither
  vendored in the reference nor installed, and the reference has no tests.
  ``remap_bilinear(mode="exact")`` is exact bilinear interpolation with
  replicate border on the unquantised float32 coordinates; ``mode="cv2"``
  follows OpenCV's published algorithm (1/32-pixel coordinate quantisation,
  table weights) and cannot be verified here.

Reduction-order convention: wherever the reference calls a library reduction
whose association order is implementation defined (torch ``sum``/``mean`` on
float32), the oracle accumulates in float64 and rounds once.  The HIP kernels
do the same, so HIP == oracle bit-for-bit there, and both sit within a couple
of float32 ulps of what torch produced for the goldens.  One exception: the
attention aggregation A1 (the only reduction that reads a bandwidth-relevant
amount of data, in the model's float16 on the model's GPU) accumulates in
float32 like torch's GPU reductions, in a fixed order written down in
``_row_sums_f32_tree`` / ``_head_sums_f32``.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import numpy as np

F32 = np.float32
F64 = np.float64

NUM_IMAGE_TOKENS = 576  # AGW/attention_extraction/llava.py:350 (24 x 24 patches)
GRID = 24


# ---------------------------------------------------------------------------
# A1 / A2  attention aggregation
# ---------------------------------------------------------------------------

def _round_to(x: np.ndarray, dtype) -> np.ndarray:
    return np.asarray(x).astype(dtype)


_XOR = {o: np.arange(64) ^ o for o in (32, 16, 8, 4, 2, 1)}


def _row_sums_f32_tree(x: np.ndarray) -> np.ndarray:
    """Sum over the last axis of a float32 array in the path's FIXED float32 order (every add rounded to float32):
    token t belongs to lane l = (t mod 256) // 4 of 64 lanes; a lane adds its tokens four at a time,
    ``(x[t] + x[t+1]) + (x[t+2] + x[t+3])``, the quads of successive 256-token blocks one after the other onto a
    running sum that starts at +0; the 64 lane sums are then combined by a butterfly, ``s[l] += s[l ^ o]`` for
    o = 32, 16, 8, 4, 2, 1.  Tokens past the end count as +0.  (This is what a 64-lane wavefront computes when every
    lane loads four consecutive elements; it is a legitimate instance of the implementation-defined order of the
    reference's ``a.sum(-1)`` on the GPU, AGW/attention_extraction/llava.py:392, which accumulates in float32.)"""
    x = np.asarray(x, dtype=F32)
    n = x.shape[-1]
    nb = max(1, -(-n // 256))
    xp = np.zeros(x.shape[:-1] + (nb * 256,), dtype=F32)
    xp[..., :n] = x
    q = xp.reshape(x.shape[:-1] + (nb, 64, 4))
    quad = (q[..., 0] + q[..., 1]) + (q[..., 2] + q[..., 3])          # float32 adds
    s = np.zeros(x.shape[:-1] + (64,), dtype=F32)
    for i in range(nb):
        s = s + quad[..., i, :]
    for o in (32, 16, 8, 4, 2, 1):
        s = s + s[..., _XOR[o]]
    return s[..., 0]


def _head_sums_f32(r: np.ndarray) -> np.ndarray:
    """Sum over heads (axis 0) of float32 [heads, ntok] in the path's fixed order: four partial sums P_w over the heads
    h = w, w+4, w+8, ... (ascending, starting at +0), combined as ((0 + P_0) + P_1) + P_2) + P_3."""
    r = np.asarray(r, dtype=F32)
    m = np.zeros(r.shape[1:], dtype=F32)
    for w in range(4):
        pw = np.zeros(r.shape[1:], dtype=F32)
        for h in range(w, r.shape[0], 4):
            pw = pw + r[h]
        m = m + pw
    return m


def attn_reduce_step(attn: np.ndarray, starts: Sequence[int], ends: Sequence[int]) -> np.ndarray:
    """One generation step of ``BatchMaskHookLogger._process_attention``.

    Follows AGW/attention_extraction/llava.py:385-396: for each sample ``b``
    take the last query row ``attn[b, :, -1, st:ed]`` (``ed`` clipped to the kv
    length), divide every head by ``(row sum + 1e-12)`` and average over heads.
    Arithmetic dtype = dtype of ``attn`` (float32 or float16): every reference op boundary rounds to it.  The two
    reductions accumulate in FLOAT32 -- as torch's GPU ``sum`` / ``mean`` do for float16 and float32 inputs alike
    (llava.py:392-394 run on the model's device) -- in the fixed orders of ``_row_sums_f32_tree`` / ``_head_sums_f32``.

    attn: [B, heads, q, kv]  ->  [B, ntok]
    """
    attn = np.asarray(attn)
    dt = attn.dtype
    B, heads, _, kv = attn.shape
    out = []
    for b in range(B):
        st = int(starts[b])
        ed = min(int(ends[b]), kv)
        a = attn[b, :, -1, st:ed]                                   # [heads, ntok]
        s = _row_sums_f32_tree(a.astype(F32))[:, None].astype(dt)   # row sum (float32 accumulate), rounded to dt
        s = (s + dt.type(1e-12)).astype(dt)                         # 1e-12 underflows to 0 in fp16
        with np.errstate(divide="ignore", invalid="ignore"):
            r = (a.astype(F32) / s.astype(F32)).astype(dt)
        m = _head_sums_f32(r.astype(F32)).astype(dt)                # torch mean = sum / N
        out.append((m.astype(F32) / F32(dt.type(heads))).astype(dt))
    return np.stack(out, axis=0)


def attn_finalize(step_attentions: Sequence[np.ndarray], batch_size: Optional[int] = None,
                  dtype=F32) -> np.ndarray:
    """``BatchMaskHookLogger.finalize_batch`` (llava.py:401-411).

    Mean over generation steps of the per-step ``[B, ntok]`` maps.  With no
    captured step the reference returns a uniform ``1/576`` map per sample
    (flat, length 576 -- llava.py:404-408).
    """
    if len(step_attentions) == 0:
        return np.full((int(batch_size), NUM_IMAGE_TOKENS), F32(1.0) / F32(NUM_IMAGE_TOKENS), dtype=F32)
    st = np.stack([np.asarray(s) for s in step_attentions], axis=0)   # [T, B, ntok]
    dt = st.dtype
    acc = st.astype(F64).sum(axis=0).astype(dt)
    return (acc / dt.type(st.shape[0])).astype(dt)


def attn_reduce_stack(rows: np.ndarray, starts: Sequence[int], ntok: int = NUM_IMAGE_TOKENS) -> np.ndarray:
    """Fused A1+A2 over a captured stack of last-query rows.

    rows: [T, B, heads, kv] (the ``[:, :, -1, :]`` rows of every step).
    Equivalent to ``attn_finalize([attn_reduce_step(step) for step in rows])``.
    """
    T = rows.shape[0]
    steps = [attn_reduce_step(rows[t][:, :, None, :], starts, [int(s) + ntok for s in starts]) for t in range(T)]
    return attn_finalize(steps)


def attn_probe_last_row(q: np.ndarray, k: np.ndarray, scaling: float,
                        kv_begin: Optional[Sequence[int]] = None) -> np.ndarray:
    """Last query row of HF's eager attention, the tensor the reference's hook slices
    (AGW/attention_extraction/llava.py:391 reads ``attn_weights[b, :, -1, st:ed]``; the weights come from
    ``register_hook_and_patch`` forcing ``output_attentions=True``, llava.py:422-438).

    Third-party arithmetic, not vendored: transformers (pinned 4.37.2 in the reference's attwarp.yaml:26)
    ``eager_attention_forward``: ``softmax(matmul(q, k^T) * scaling + mask, dtype=float32).to(q.dtype)``
    (4.37.2 writes ``/ sqrt(head_dim)`` instead of ``* scaling``; equal for power-of-4 head_dim, within one
    float32 ulp before the rounding to the model dtype otherwise).  Dtype transitions kept; the dot
    product and the softmax denominator are accumulated in float64 and rounded once.

    q [B,H,D] (last query, post-RoPE), k [B,Hkv,kv,D] -> probabilities [B,H,kv] in q's dtype.
    kv_begin[b]: first attended key (left padding); earlier keys carry the mask's finfo.min -> 0.
    """
    q, k = np.asarray(q), np.asarray(k)
    dt = q.dtype
    B, H, D = q.shape
    Hkv, kv = k.shape[1], k.shape[2]
    kk = np.repeat(k, H // Hkv, axis=1)                                    # repeat_kv
    w = np.einsum("bhd,bhjd->bhj", q.astype(F64), kk.astype(F64)).astype(dt)
    w = (w.astype(F32) * F32(scaling)).astype(dt)
    x = w.astype(F32)
    out = np.zeros((B, H, kv), dtype=dt)
    for b in range(B):
        kb = 0 if kv_begin is None else int(kv_begin[b])
        xv = x[b, :, kb:]
        m = xv.max(axis=-1, keepdims=True)
        e = np.exp((xv - m).astype(F32).astype(F64)).astype(F32)
        den = e.astype(F64).sum(axis=-1, keepdims=True).astype(F32)
        out[b, :, kb:] = (e / den).astype(dt)
    return out


def attn_probe_step(q: np.ndarray, k: np.ndarray, starts: Sequence[int], ntok: int, scaling: float,
                    kv_begin: Optional[Sequence[int]] = None) -> np.ndarray:
    """Probed generation step = ``attn_reduce_step`` of ``attn_probe_last_row``: [B, ntok]."""
    p = attn_probe_last_row(q, k, scaling, kv_begin)
    return attn_reduce_step(p[:, :, None, :], starts, [int(s) + ntok for s in starts])


# ---------------------------------------------------------------------------
# A3  mask post-processing (24 x 24)
# ---------------------------------------------------------------------------

def revise_mask(mask: np.ndarray, kernel_size: int = 3, enhance_coe: float = 10.0) -> np.ndarray:
    """``revise_mask`` = normalize("min") -> enhance -> k x k box filter.

    Follows llava.py:207-238.  mask: [B, 24, 24] or [24, 24] float32.
    * normalize "min": (m - min) / (max - min)                         (:210-211)
    * enhance: m - mean; / std (unbiased, torch default); * coe; sigmoid;
      clamp to [0, 1]                                                 (:215-221)
    * Conv2d(1, 1, k, padding=(k-1)/2, padding_mode="replicate") with all
      weights 1/k**2                                                  (:229-233)
    Internally float64, rounded to float32 at each reference op boundary that
    is a reduction or transcendental; elementwise float32 ops are exact.
    """
    assert kernel_size % 2 == 1
    m = np.asarray(mask, dtype=F32)
    single = m.ndim == 2
    if single:
        m = m[None]
    B = m.shape[0]
    out = np.empty_like(m)
    pad = (kernel_size - 1) // 2
    wgt = F32(1.0) / F32(kernel_size ** 2)            # ones_like(weight) / kernel_size**2 in float32
    for b in range(B):
        x = m[b]
        mn, mx = x.min(), x.max()
        x = ((x - mn) / (mx - mn)).astype(F32)
        n = x.size
        mean = F32(x.astype(F64).sum() / n)
        x = (x - mean).astype(F32)
        # unbiased std of the mean-subtracted map (torch recomputes the mean internally)
        xd = x.astype(F64)
        mu2 = xd.sum() / n
        var = ((xd - mu2) ** 2).sum() / (n - 1)
        std = F32(math.sqrt(var))
        x = (x / std).astype(F32)
        x = (x * F32(enhance_coe)).astype(F32)
        x = (1.0 / (1.0 + np.exp(-x.astype(F64)))).astype(F32)
        x = np.clip(x, F32(0), F32(1))
        xp = np.pad(x, pad, mode="edge").astype(F64)
        acc = np.zeros_like(x, dtype=F64)
        for dy in range(kernel_size):
            for dx in range(kernel_size):
                acc += (xp[dy:dy + x.shape[0], dx:dx + x.shape[1]].astype(F32) * wgt).astype(F64)
        out[b] = acc.astype(F32)
    return out[0] if single else out


# ---------------------------------------------------------------------------
# A4  mask -> uint8 -> PIL LANCZOS up-sample
# ---------------------------------------------------------------------------

def mask_to_u8(mask: np.ndarray) -> np.ndarray:
    """``T.ToPILImage()`` on a float tensor: ``pic.mul(255).byte()`` -- a
    truncating cast (llava.py:192-193, :243).  Input values are in [0, 1] -- or NaN: a CONSTANT attention map (the driver's
    OOM fallback ``torch.ones(24, 24) / 576``, main_batched.py:231) makes revise_mask divide 0 by 0 (llava.py:207-213), and
    the float -> uint8 cast of NaN is implementation defined.  The reference's CPU path yields 0 for it (pinned by
    tests/golden/main_batched_loop.npz, generated by importing the reference: mask of the constant map -> all zeros), stated
    here explicitly instead of left to numpy's cast (which warns and happens to agree on x86)."""
    v = (np.asarray(mask, dtype=F32) * F32(255.0)).astype(F32)
    nan = np.isnan(v)
    return np.where(nan, 0, np.trunc(np.where(nan, F32(0), v)).astype(np.int64)).astype(np.uint8)


_PIL_PRECISION_BITS = 32 - 8 - 2   # Pillow src/libImaging/Resample.c


def _lanczos(x: float) -> float:
    def sinc(t: float) -> float:
        if t == 0.0:
            return 1.0
        t = t * math.pi
        return math.sin(t) / t
    if -3.0 <= x < 3.0:
        return sinc(x) * sinc(x / 3.0)
    return 0.0


def _bicubic(x: float) -> float:
    """Pillow's bicubic_filter (a = -0.5), support 2."""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


_PIL_FILTERS = {"lanczos": (_lanczos, 3.0), "bicubic": (_bicubic, 2.0)}


def pil_resample_coeffs(in_size: int, out_size: int, filt: str = "lanczos") -> Tuple[np.ndarray, np.ndarray, int]:
    """Pillow's ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` (src/libImaging/Resample.c) over the full
    box ``[0, in_size)`` for the LANCZOS (support 3) or BICUBIC (support 2) filter.

    Returns (bounds[out, 2] = (xmin, count), kk[out, ksize] int32, ksize).
    Pillow is the third-party dependency behind ``invtrans`` (llava.py:195-196) and behind the CLIP
    image processor's resize; version used for the goldens: Pillow 12.2.0.
    """
    fn, support0 = _PIL_FILTERS[filt]
    scale = float(in_size) / float(out_size)
    filterscale = max(scale, 1.0)
    support = support0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [fn((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            if v < 0:
                kk[xx, x] = int(-0.5 + v * (1 << _PIL_PRECISION_BITS))
            else:
                kk[xx, x] = int(0.5 + v * (1 << _PIL_PRECISION_BITS))
        bounds[xx, 0] = xmin
        bounds[xx, 1] = xmax
    return bounds, kk, ksize


def pil_lanczos_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    return pil_resample_coeffs(in_size, out_size, "lanczos")


def _pil_resample_axis_u8(src: np.ndarray, out_size: int, filt: str = "lanczos") -> np.ndarray:
    """One 8-bit pass of Pillow's separable resampler along the LAST axis."""
    in_size = src.shape[-1]
    bounds, kk, ksize = pil_resample_coeffs(in_size, out_size, filt)
    out = np.empty(src.shape[:-1] + (out_size,), dtype=np.uint8)
    s = src.astype(np.int64)
    for xx in range(out_size):
        xmin, cnt = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.full(src.shape[:-1], 1 << (_PIL_PRECISION_BITS - 1), dtype=np.int64)
        for x in range(cnt):
            acc += s[..., xmin + x] * int(kk[xx, x])
        out[..., xx] = np.clip(acc >> _PIL_PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def pil_resize_u8(img: np.ndarray, out_w: int, out_h: int, filt: str = "lanczos") -> np.ndarray:
    """``PIL.Image.resize((out_w, out_h), <filter>)`` for an 8-bit image, mode "L" ([H,W]) or
    multi-band ([H,W,C]; Pillow resamples every band with the same integer arithmetic).

    Two passes, horizontal first then vertical, uint8 intermediate (``ImagingResampleInner``).
    A pass is skipped when that axis keeps its size."""
    img = np.asarray(img, dtype=np.uint8)
    squeeze = img.ndim == 2
    cur = img[:, :, None] if squeeze else img
    h, w, _ = cur.shape
    if out_w != w:
        cur = _pil_resample_axis_u8(cur.transpose(0, 2, 1), out_w, filt).transpose(0, 2, 1)
    if out_h != h:
        cur = _pil_resample_axis_u8(cur.transpose(1, 2, 0), out_h, filt).transpose(2, 0, 1)
    cur = np.ascontiguousarray(cur)
    return cur[:, :, 0] if squeeze else cur


def lanczos_resize_u8(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """``PIL.Image.resize((out_w, out_h), LANCZOS)`` for a mode-"L" image (A4, llava.py:195-196,253)."""
    return pil_resize_u8(img, out_w, out_h, "lanczos")


# ---------------------------------------------------------------------------
# "next" row 3: warped uint8 image -> CLIP-ready tensor (what LLaVA-1.5 feeds its vision tower)
# ---------------------------------------------------------------------------

OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def clip_preprocess(img_rgb_u8: np.ndarray, size: int = 336) -> np.ndarray:
    """HF ``CLIPImageProcessor`` (PIL backend) as LLaVA-1.5 configures it, for an RGB uint8 image [H,W,3]:
    resize so that the SHORTER edge is ``size`` (PIL BICUBIC, long edge int(size*long/short)), center crop
    ``size`` x ``size``, rescale ``float32(float64(u8) * (1/255))``, normalize ``(x - mean32) / std32`` in
    float32, channels first.  Returns float32 [3,size,size] (LLaVA casts to float16 afterwards:
    AGW/attention_extraction/functions.py:270-271; the evaluation path re-reads the warped PNG,
    AGW/evaluate_accuracy.py:157-158).  Pinned to transformers 5.15's PIL backend + Pillow 12.2.0 goldens
    (the reference pins transformers 4.37.2, whose slow processor does the same arithmetic)."""
    img = np.asarray(img_rgb_u8, dtype=np.uint8)
    h, w, c = img.shape
    short, long_ = (h, w) if h <= w else (w, h)
    new_short, new_long = size, int(size * long_ / short)
    nh, nw = (new_short, new_long) if h <= w else (new_long, new_short)
    r = pil_resize_u8(img, nw, nh, "bicubic")
    top, left = (nh - size) // 2, (nw - size) // 2
    r = r[top:top + size, left:left + size]
    x = (r.astype(F64) * (1 / 255)).astype(F32)
    mean = np.array(OPENAI_CLIP_MEAN, dtype=F32)
    std = np.array(OPENAI_CLIP_STD, dtype=F32)
    x = ((x - mean).astype(F32) / std).astype(F32)
    return np.ascontiguousarray(x.transpose(2, 0, 1))


def expand2square(img_rgb_u8: np.ndarray, background=None) -> np.ndarray:
    """LLaVA-1.5's ``image_aspect_ratio == "pad"`` step in front of the CLIP processor (``process_images`` ->
    ``expand2square(image, tuple(int(x * 255) for x in image_processor.image_mean))``; external LLaVA checkout,
    ``llava/mm_utils.py``, NOT part of /root/reference -- restated from the published algorithm, the reference only
    reaches it through ``process_images``, AGW/attention_extraction/functions.py:262-271 and
    AGW/evaluate_accuracy.py:157-158).  [H,W,C] uint8 -> [max(H,W), max(H,W), C]: the image pasted at
    ``(long - short) // 2`` along the short axis on a canvas of the background colour.  A square image (the 500 x 500
    warps of both reference drivers) is returned unchanged."""
    img = np.asarray(img_rgb_u8, dtype=np.uint8)
    h, w, c = img.shape
    if background is None:
        background = tuple(int(x * 255) for x in OPENAI_CLIP_MEAN)
    if h == w:
        return img
    n = max(h, w)
    out = np.empty((n, n, c), np.uint8)
    out[...] = np.asarray(background[:c], np.uint8)
    if w > h:
        o = (w - h) // 2
        out[o:o + h] = img
    else:
        o = (h - w) // 2
        out[:, o:o + w] = img
    return out


# ---------------------------------------------------------------------------
# A5  24 x 24 adaptive average pool
# ---------------------------------------------------------------------------

def adaptive_windows(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """torch adaptive-pool windows: start = floor(i*L/n), end = ceil((i+1)*L/n)
    (MN/checkpoint_utils.py:104-107 restates the same rule)."""
    i = np.arange(out_size, dtype=np.int64)
    starts = (i * in_size) // out_size
    ends = ((i + 1) * in_size + out_size - 1) // out_size
    return starts, ends


def adaptive_avg_pool24(A: np.ndarray, out_hw: Tuple[int, int] = (GRID, GRID)) -> np.ndarray:
    """``F.adaptive_avg_pool2d(A_full, (24, 24))`` (MN/trainer.py:197,433,465).

    A: [B, 1, H, W] float32 -> [B, 1, 24, 24] float32 (float64 accumulate,
    one rounding, then one float32 divide by the window size as ATen does)."""
    A = np.asarray(A, dtype=F32)
    B, C, H, W = A.shape
    oh, ow = out_hw
    ys, ye = adaptive_windows(H, oh)
    xs, xe = adaptive_windows(W, ow)
    out = np.empty((B, C, oh, ow), dtype=F32)
    Ad = A.astype(F64)
    for i in range(oh):
        for j in range(ow):
            win = Ad[:, :, ys[i]:ye[i], xs[j]:xe[j]]
            cnt = (ye[i] - ys[i]) * (xe[j] - xs[j])
            out[:, :, i, j] = (win.sum(axis=(2, 3)).astype(F32) / F32(cnt)).astype(F32)
    return out


# ---------------------------------------------------------------------------
# A6  marginals, A7 safe softmax
# ---------------------------------------------------------------------------

def gt_marginals(A: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """``gt_marginals`` (MN/checkpoint_utils.py:43-51).  A: [B,1,H,W] ->
    (px [B,W], py [B,H]); clamp >= 0, sum rows / cols, normalise by
    ``sum.clamp_min(1e-6)``."""
    A = np.asarray(A, dtype=F32)
    Apos = np.maximum(A, F32(0))[:, 0].astype(F64)
    mx = Apos.sum(axis=1).astype(F32)            # (B, W)
    my = Apos.sum(axis=2).astype(F32)            # (B, H)
    sx = np.maximum(mx.astype(F64).sum(axis=1, keepdims=True).astype(F32), F32(1e-6))
    sy = np.maximum(my.astype(F64).sum(axis=1, keepdims=True).astype(F32), F32(1e-6))
    return (mx / sx).astype(F32), (my / sy).astype(F32)


def safe_softmax(logits: np.ndarray, eps: float = 1e-6) -> np.ndarray:
    """``safe_softmax`` over dim=1 (MN/model.py:8-14): nan_to_num(->0) ->
    subtract amax -> softmax -> nan_to_num -> / sum.clamp_min(eps)."""
    x = np.asarray(logits, dtype=F32)
    x = np.nan_to_num(x, nan=0.0, posinf=0.0, neginf=0.0).astype(F32)
    x = (x - x.max(axis=1, keepdims=True)).astype(F32)
    # F.softmax subtracts the row max again (it is 0 here) and exponentiates.
    x = (x - x.max(axis=1, keepdims=True)).astype(F32)
    e = np.exp(x.astype(F64)).astype(F32)
    s = e.astype(F64).sum(axis=1, keepdims=True).astype(F32)
    p = (e / s).astype(F32)
    p = np.nan_to_num(p, nan=0.0, posinf=0.0, neginf=0.0).astype(F32)
    d = np.maximum(p.astype(F64).sum(axis=1, keepdims=True).astype(F32), F32(eps))
    return (p / d).astype(F32)


# ---------------------------------------------------------------------------
# "next" row 1: MarginalNet's text pooling and FiLM + axis means (MN/model.py:73-88)
# ---------------------------------------------------------------------------

def masked_token_mean(tok: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """MN/model.py:77-78: ``denom = mask.sum(1).clamp_min(1); t = (tok * mask).sum(1) / denom``.
    tok [B,Lt,D] float32, mask [B,Lt] (the reference's [B,Lt,1]) -> [B,D] float32.
    Products rounded to float32, the token sum accumulated in float64 and rounded once."""
    tok = np.asarray(tok, F32)
    m = np.asarray(mask, F32).reshape(tok.shape[0], tok.shape[1], 1)
    denom = np.maximum(m.astype(F64).sum(axis=1).astype(F32), F32(1.0))              # [B,1]
    prod = (tok * m).astype(F32)
    return (prod.astype(F64).sum(axis=1).astype(F32) / denom).astype(F32)


def film_axis_means(v: np.ndarray, gamma_beta: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """MN/model.py:80-88: ``gamma, beta = film(t).chunk(2, 1); v = gamma*v + beta; vx = v.mean(2); vy = v.mean(3)``.
    v [B,Ch,H,W] float32, gamma_beta [B,2*Ch] -> (vx [B,Ch,W], vy [B,Ch,H]).
    ``gamma*v`` and ``+ beta`` are two float32 roundings (two torch kernels); torch's mean = sum / N with an
    implementation-defined float32 sum order: accumulated in float64 and rounded once here."""
    v = np.asarray(v, F32)
    gb = np.asarray(gamma_beta, F32)
    B, Ch, H, W = v.shape
    gamma, beta = gb[:, :Ch, None, None], gb[:, Ch:, None, None]
    f = ((gamma * v).astype(F32) + beta).astype(F32)
    vx = (f.astype(F64).sum(axis=2).astype(F32) / F32(H)).astype(F32)
    vy = (f.astype(F64).sum(axis=3).astype(F32) / F32(W)).astype(F32)
    return vx, vy


# ---------------------------------------------------------------------------
# A8  right-inverse PDF up-sample
# ---------------------------------------------------------------------------

def pooling_matrix(L_out: int, L_in: int, dtype=F32) -> np.ndarray:
    """Adaptive-avg-pool1d matrix A [L_out, L_in] (MN/checkpoint_utils.py:104-113):
    row k = 1/(e-s) on [s, e)."""
    starts, ends = adaptive_windows(L_in, L_out)
    A = np.zeros((L_out, L_in), dtype=dtype)
    for k in range(L_out):
        s, e = int(starts[k]), int(ends[k])
        A[k, s:e] = dtype(1.0) / dtype(max(e - s, 1))
    return A


def right_inverse_core(L_out: int, L_in: int, eps: float = 1e-8) -> np.ndarray:
    """(A A^T + eps I)^-1 as float64 [L_out, L_out] built from the float32
    pooling matrix -- the tiny system the reference solves by LU per call
    (MN/checkpoint_utils.py:116-120)."""
    A = pooling_matrix(L_out, L_in, F32).astype(F64)
    # BLAS-free (hence reproducible) Gram matrix, then rounded to the float32 the reference forms it in
    AAT = (A[:, None, :] * A[None, :, :]).sum(axis=2).astype(F32)
    if eps > 0:
        AAT = (AAT + (F32(eps) * np.eye(L_out, dtype=F32))).astype(F32)
    AAT = AAT.astype(F64)
    return np.linalg.inv(AAT)


def upsample_pdf_right_inverse(y: np.ndarray, target_len: int, eps: float = 1e-8) -> np.ndarray:
    """``upsample_pdf_right_inverse`` (MN/checkpoint_utils.py:64-131):
    x_hat = A^T (A A^T + eps I)^-1 y for 1-D / 2-D / 3-D ``y``.

    Restatement: tmp = solve(...) is evaluated as ``inv @ y`` in float64 and
    rounded to float32; x_hat[l] = sum_k A[k, l] * tmp[k] where column l of A
    has one or two non-zeros (added in ascending k, float32)."""
    y = np.asarray(y, dtype=F32)
    if y.ndim == 1:
        yN = y[None]
    elif y.ndim == 2:
        yN = y
    elif y.ndim == 3:
        yN = y.reshape(-1, y.shape[-1])
    else:
        raise ValueError(f"upsample_pdf_right_inverse expects 1D/2D/3D y; got shape {tuple(y.shape)}")
    L_out = yN.shape[1]
    L_in = int(target_len)
    inv = right_inverse_core(L_out, L_in, eps)
    acc = np.zeros((yN.shape[0], L_out), dtype=F64)              # tmp[n,k] = sum_j y[n,j]*inv[k,j], j ascending
    for j in range(L_out):
        acc = acc + yN[:, j:j + 1].astype(F64) * inv[:, j][None, :]
    tmp = acc.astype(F32)                                        # [N, L_out]
    A = pooling_matrix(L_out, L_in, F32)
    x = np.zeros((yN.shape[0], L_in), dtype=F32)
    for k in range(L_out):
        nz = np.nonzero(A[k])[0]
        x[:, nz] = (x[:, nz] + (tmp[:, k:k + 1] * A[k, nz][None, :]).astype(F32)).astype(F32)
    if y.ndim == 1:
        return x[0]
    if y.ndim == 3:
        return x.reshape(y.shape[0], y.shape[1], L_in)
    return x


# ---------------------------------------------------------------------------
# A9 / A10  CDFs
# ---------------------------------------------------------------------------

def cdf_from_density(p: np.ndarray) -> np.ndarray:
    """``cdf_from_density`` (MN/checkpoint_utils.py:30-41): clamp >= 0,
    nan_to_num(->0), / sum.clamp_min(1e-6), cumsum (float64 accumulate, each
    prefix rounded to float32 -- what torch-CPU cumsum does), last = 1."""
    p = np.asarray(p, dtype=F32)
    p = np.maximum(p, F32(0))            # clamp_min(0) keeps NaN
    p = np.where(np.isnan(p), F32(0), p)
    p = np.nan_to_num(p, nan=0.0, posinf=0.0, neginf=0.0).astype(F32)
    denom = np.maximum(p.astype(F64).sum(axis=1, keepdims=True).astype(F32), F32(1e-6))
    p = (p / denom).astype(F32)
    Fp = np.cumsum(p.astype(F64), axis=1).astype(F32)
    Fp[:, -1] = F32(1.0)
    return Fp


def make_strictly_increasing(Fcdf: np.ndarray, eps: float = 1e-4) -> np.ndarray:
    """``_make_strictly_increasing`` (MN/checkpoint_utils.py:17-28)."""
    Fc = np.asarray(Fcdf, dtype=F32)
    Fc = np.nan_to_num(Fc, nan=0.0, posinf=1.0, neginf=0.0).astype(F32)
    nd = np.maximum.accumulate(Fc, axis=1)
    B, N = nd.shape
    min_step = F32(eps / max(N, 1))
    d = np.maximum((nd[:, 1:] - nd[:, :-1]).astype(F32), min_step)
    cs = np.cumsum(d.astype(F64), axis=1).astype(F32)
    fix = np.concatenate([nd[:, :1], (nd[:, :1] + cs).astype(F32)], axis=1)
    last = np.maximum(fix[:, -1:], F32(1e-6))
    fix = np.clip((fix / last).astype(F32), F32(0), F32(1))
    fix[:, -1] = F32(1.0)
    return fix


def interpolate_linear_align_corners(x: np.ndarray, size: int) -> np.ndarray:
    """``F.interpolate(mode="linear", align_corners=True)`` along the last dim
    (ATen upsample_linear1d: float32 scale = (in-1)/(out-1), src = scale*i,
    i0 = floor, lam = src - i0, out = (1-lam)*x[i0] + lam*x[i1])."""
    x = np.asarray(x, dtype=F32)
    n = x.shape[-1]
    if size == 1:
        scale = F32(0)
    else:
        scale = F32(n - 1) / F32(size - 1)
    i = np.arange(size, dtype=F32)
    src = (scale * i).astype(F32)
    i0 = np.minimum(src.astype(np.int64), n - 1)
    i1 = np.minimum(i0 + 1, n - 1)
    lam1 = (src - i0.astype(F32)).astype(F32)
    lam0 = (F32(1) - lam1).astype(F32)
    return ((lam0 * x[..., i0]).astype(F32) + (lam1 * x[..., i1]).astype(F32)).astype(F32)


def resample_cdf(Fcdf: np.ndarray, target_len: int) -> np.ndarray:
    """``resample_cdf`` (MN/checkpoint_utils.py:53-62)."""
    Fc = make_strictly_increasing(np.asarray(Fcdf, dtype=F32))
    up = interpolate_linear_align_corners(Fc, int(target_len))
    return make_strictly_increasing(up)


# ---------------------------------------------------------------------------
# A11  maps from CDFs (torch variant)
# ---------------------------------------------------------------------------

def axis_map_from_cdf(F_b: np.ndarray, n_out: int) -> np.ndarray:
    """One axis of the grid construction in ``warp_from_cdf_torch``
    (MN/checkpoint_utils.py:167-193).  F_b: [L] float32 CDF -> float32 [n_out]
    source coordinate for every output index."""
    F_b = np.asarray(F_b, dtype=F32).reshape(-1)
    L = F_b.shape[0]
    orig = np.arange(L, dtype=F32)
    new_fwd = np.concatenate(([0.0], F_b)) * float(n_out)          # float64
    orig_fwd = np.concatenate(([0.0], orig + 1.0))
    new_fwd[-1] = n_out
    if np.any(np.diff(new_fwd) <= 0):
        new_fwd += (1e-4 / max(n_out, 1)) * np.arange(new_fwd.size, dtype=F32)
    target = np.arange(n_out, dtype=F32)
    return np.interp(target, new_fwd, orig_fwd).astype(F32)


def maps_from_cdf(Fx: np.ndarray, Fy: np.ndarray, out_size: Optional[Tuple[int, int]] = None
                  ) -> Tuple[np.ndarray, np.ndarray]:
    """Batched 1-D maps: Fx [B,W], Fy [B,H] -> map_x [B,W_out], map_y [B,H_out].
    The reference expands these with ``np.meshgrid(indexing="xy")`` into dense
    [H_out, W_out] float32 maps (:191-193); the dense maps carry no more
    information than the two vectors."""
    Fx = np.asarray(Fx, dtype=F32)
    Fy = np.asarray(Fy, dtype=F32)
    B, W = Fx.shape
    H = Fy.shape[1]
    H_out, W_out = (H, W) if out_size is None else out_size
    mx = np.stack([axis_map_from_cdf(Fx[b], W_out) for b in range(B)])
    my = np.stack([axis_map_from_cdf(Fy[b], H_out) for b in range(B)])
    return mx, my


# ---------------------------------------------------------------------------
# A13  maps from a full-resolution attention map (numpy / float64 variant)
# ---------------------------------------------------------------------------

TRANSFORMS = ("identity", "square", "sqrt", "exp", "log")
EPSILON = 1e-9          # AGW/new_method.py:193
BASE_ATTENTION = 1e-9   # AGW/new_method.py:194


def _transform(x: np.ndarray, name: str, exp_scale: float, exp_divisor: float) -> np.ndarray:
    """AGW/new_method.py:134-179."""
    if name == "identity":
        return x
    if name == "square":
        return x ** 2
    if name == "sqrt":
        return np.sqrt(np.maximum(x, 0))
    if name == "exp":
        return np.exp(exp_scale * x) / exp_divisor
    if name == "log":
        return np.log(x + 1e-5)
    raise ValueError(name)


def _inverse(x: np.ndarray, name: str, exp_scale: float, exp_divisor: float) -> np.ndarray:
    """AGW/new_method.py:138-179 (the inverse of each transform)."""
    if name == "identity":
        return x
    if name == "square":
        return np.sqrt(np.maximum(x, 0))
    if name == "sqrt":
        return x ** 2
    if name == "exp":
        return np.log(np.maximum(x * exp_divisor, 1e-9)) / exp_scale
    if name == "log":
        return np.exp(x) - 1e-5
    raise ValueError(name)


def maps_from_attention(att_map: np.ndarray, new_width: int, new_height: int,
                        transform: str = "identity", exp_scale: float = 1.0,
                        exp_divisor: float = 1.0, apply_inverse: bool = False
                        ) -> Tuple[np.ndarray, np.ndarray]:
    """Grid construction of ``warp_image_by_attention`` (AGW/new_method.py:206-265)
    returning the two 1-D float32 maps (the reference meshgrids them :263-265).

    Unknown transform names fall back to identity (``set_transform_function``
    :399-401)."""
    if transform not in TRANSFORMS:
        transform = "identity"
    h, w = att_map.shape[:2]
    a = att_map.astype(F64)
    a = np.maximum(a, 0)
    with np.errstate(all="ignore"):
        a = _transform(a, transform, exp_scale, exp_divisor)
    biased = a + BASE_ATTENTION
    prof_x = np.sum(biased, axis=0)
    prof_y = np.sum(biased, axis=1)
    if apply_inverse:
        with np.errstate(all="ignore"):
            prof_x = _inverse(prof_x - BASE_ATTENTION * h, transform, exp_scale, exp_divisor)
            prof_y = _inverse(prof_y - BASE_ATTENTION * w, transform, exp_scale, exp_divisor)
        prof_x = prof_x + BASE_ATTENTION * h
        prof_y = prof_y + BASE_ATTENTION * w
    total_x = np.sum(prof_x)
    total_y = np.sum(prof_y)
    if total_x < EPSILON or total_y < EPSILON:
        prof_x = np.ones(w, dtype=F64)
        prof_y = np.ones(h, dtype=F64)
        total_x = w * (np.mean(biased) * h)
        total_y = h * (np.mean(biased) * w)
        total_x = max(total_x, EPSILON)
        total_y = max(total_y, EPSILON)
    cum_x = np.cumsum(prof_x)
    cum_y = np.cumsum(prof_y)
    with np.errstate(all="ignore"):        # NaN attention propagates as in the reference
        fwd_x = np.concatenate(([0], cum_x / total_x)) * new_width
        fwd_y = np.concatenate(([0], cum_y / total_y)) * new_height
    orig_x = np.concatenate(([0], np.arange(w) + 1))
    orig_y = np.concatenate(([0], np.arange(h) + 1))
    fwd_x[-1] = new_width
    fwd_y[-1] = new_height
    map_x = np.interp(np.arange(new_width), fwd_x, orig_x)
    map_y = np.interp(np.arange(new_height), fwd_y, orig_y)
    return map_x.astype(F32), map_y.astype(F32)


# ---------------------------------------------------------------------------
# A12  bilinear resample with replicate border
# ---------------------------------------------------------------------------

def _axis_taps(m: np.ndarray, size: int):
    """float32 coordinate -> (i0 clamped, i1 clamped, frac float32).  The
    fraction is taken BEFORE clamping (replicate border: both taps collapse on
    the edge pixel, the weight no longer matters)."""
    m = np.asarray(m, dtype=F32)
    # the coordinate is clamped to [-1, size] first, NaN counting as -1 (C fmaxf semantics): non-finite and huge
    # coordinates land on an edge pixel and never produce a NaN weight
    m = np.where(np.isnan(m), F32(-1), m).astype(F32)
    m = np.minimum(np.maximum(m, F32(-1)), F32(size)).astype(F32)
    fl = np.floor(m)
    frac = (m - fl).astype(F32)
    i0 = fl.astype(np.int64)
    i1 = i0 + 1
    return np.clip(i0, 0, size - 1), np.clip(i1, 0, size - 1), frac


def _lerp(a: np.ndarray, b: np.ndarray, t: np.ndarray) -> np.ndarray:
    """a + t*(b - a), each operation rounded to float32 (no fused multiply-add)."""
    d = (b - a).astype(F32)
    return (a + (t * d).astype(F32)).astype(F32)


def remap_bilinear(src: np.ndarray, map_x: np.ndarray, map_y: np.ndarray, mode: str = "cv2") -> np.ndarray:
    """Stand-in for ``cv2.remap(src, meshgrid(map_x, map_y), INTER_LINEAR,
    BORDER_REPLICATE)`` (AGW/new_method.py:268-271, MN/checkpoint_utils.py:195-198).

    src: [H, W, C] (or [H, W]) float32 or uint8; map_x: [W_out], map_y: [H_out]
    float32 source coordinates.  PARITY UNPINNED (see module docstring).

    mode="exact": vertical lerp of the two source rows, then horizontal lerp,
      float32, weights = fractional parts of the unquantised coordinates.
      uint8 sources are interpolated in float32 and rounded half-to-even.
    mode="cv2": OpenCV's scheme -- coordinates rounded to 1/32 pixel
      (``cvRound(v * 32)``), four float32 table weights for float sources,
      int16 weights scaled by 2**15 with ``(sum + 2**14) >> 15`` for uint8.
    """
    src = np.asarray(src)
    squeeze = src.ndim == 2
    if squeeze:
        src = src[:, :, None]
    H, W, C = src.shape
    is_u8 = src.dtype == np.uint8
    if src.dtype == np.float64:
        return _remap_f64(src, np.asarray(map_x, F32), np.asarray(map_y, F32), mode)[:, :, 0 if squeeze else slice(None)]
    if mode == "exact":
        x0, x1, fx = _axis_taps(map_x, W)
        y0, y1, fy = _axis_taps(map_y, H)
        s = src.astype(F32)
        top = s[y0]                       # [H_out, W, C]
        bot = s[y1]
        v = _lerp(top, bot, fy[:, None, None])
        out = _lerp(v[:, x0], v[:, x1], fx[None, :, None])
        if is_u8:
            out = np.clip(np.rint(out), 0, 255).astype(np.uint8)
    elif mode == "cv2":
        out = _remap_cv2_compat(src, np.asarray(map_x, F32), np.asarray(map_y, F32))
    else:
        raise ValueError(mode)
    return out[:, :, 0] if squeeze else out


def _remap_f64(src: np.ndarray, map_x: np.ndarray, map_y: np.ndarray, mode: str) -> np.ndarray:
    """float64 images (``warp_from_cdf_torch`` hands the image to ``cv2.remap`` in its own dtype,
    MN/checkpoint_utils.py:152,195-203): OpenCV's CV_64F path keeps the float32 table weights and accumulates the four
    products in double, left to right; ``exact``: the three lerps in double with the float32 coordinate fractions."""
    H, W, C = src.shape
    if mode == "exact":
        x0, x1, fx = _axis_taps(map_x, W)
        y0, y1, fy = _axis_taps(map_y, H)
        top, bot = src[y0], src[y1]
        v = top + fy.astype(F64)[:, None, None] * (bot - top)
        a, b = v[:, x0], v[:, x1]
        return a + fx.astype(F64)[None, :, None] * (b - a)
    sx = _cv_round(map_x * F32(32)); sy = _cv_round(map_y * F32(32))
    ix, kx = sx >> 5, sx & 31
    iy, ky = sy >> 5, sy & 31
    x0 = np.clip(ix, 0, W - 1); x1 = np.clip(ix + 1, 0, W - 1)
    y0 = np.clip(iy, 0, H - 1); y1 = np.clip(iy + 1, 0, H - 1)
    tx = (kx.astype(F32) * F32(1 / 32)).astype(F32); ty = (ky.astype(F32) * F32(1 / 32)).astype(F32)
    ox = (F32(1) - tx).astype(F32); oy = (F32(1) - ty).astype(F32)
    w00 = (oy[:, None] * ox[None, :]).astype(F32).astype(F64)[..., None]
    w01 = (oy[:, None] * tx[None, :]).astype(F32).astype(F64)[..., None]
    w10 = (ty[:, None] * ox[None, :]).astype(F32).astype(F64)[..., None]
    w11 = (ty[:, None] * tx[None, :]).astype(F32).astype(F64)[..., None]
    p00 = src[y0][:, x0]; p01 = src[y0][:, x1]; p10 = src[y1][:, x0]; p11 = src[y1][:, x1]
    return ((p00 * w00 + p01 * w01) + p10 * w10) + p11 * w11


def resize_linear_cv2(img: np.ndarray, size_wh: Tuple[int, int]) -> np.ndarray:
    """``cv2.resize(img, (W_out, H_out), interpolation=cv2.INTER_LINEAR)`` (AGW/new_method.py:369) restated from
    OpenCV's published algorithm (modules/imgproc/src/resize.cpp).  PARITY UNPINNED (OpenCV is absent here).

    * ``scale = 1 / (dst / src)`` (double); ``f = float32((d + 0.5) * scale - 0.5)``, ``s = floor(f)``, ``f -= s``;
      columns: ``s < 0 -> (0, f = 0)``, ``s >= W - 1 -> (W - 1, f = 0)``; rows: indices clipped, f kept;
    * uint8: coefficients ``saturate_cast<short>(c * 2048)`` (round half to even); horizontal pass
      ``D = S[s] * a0 + S[s + 1] * a1`` in int32; vertical ``(((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2``;
    * float32: ``D = S[s] * a0 + S[s + 1] * a1``, ``out = D0 * b0 + D1 * b1``, every operation rounded to float32;
    * an exact 2 x 2 decimation is INTER_AREA: ``(s00 + s01 + s10 + s11 + 2) >> 2`` (float32: ``sum * 0.25``).
    img [H,W] or [H,W,C] uint8 / float32."""
    img = np.asarray(img)
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    if img.dtype not in (np.uint8, F32):
        raise TypeError("resize_linear_cv2: uint8 or float32")
    H, W, C = img.shape
    Wo, Ho = int(size_wh[0]), int(size_wh[1])
    is_u8 = img.dtype == np.uint8
    if W == 2 * Wo and H == 2 * Ho:
        a, b, c, d = img[0::2, 0::2], img[0::2, 1::2], img[1::2, 0::2], img[1::2, 1::2]
        if is_u8:
            out = ((a.astype(np.int64) + b + c + d + 2) >> 2).astype(np.uint8)
        else:
            out = ((((a + b).astype(F32) + c).astype(F32) + d).astype(F32) * F32(0.25)).astype(F32)
        return out[:, :, 0] if squeeze else out

    def axis(n_out, n_in, reset):
        scale = 1.0 / (float(n_out) / float(n_in))
        f = ((np.arange(n_out, dtype=F64) + 0.5) * scale - 0.5).astype(F32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(F32)).astype(F32)
        if reset:
            lo, hi = s < 0, s >= n_in - 1
            f = np.where(lo | hi, F32(0), f).astype(F32)
            s = np.where(lo, 0, np.where(hi, n_in - 1, s))
        return np.clip(s, 0, n_in - 1), np.clip(s + 1, 0, n_in - 1), f

    x0, x1, fx = axis(Wo, W, True)
    y0, y1, fy = axis(Ho, H, False)
    if is_u8:
        a0 = np.rint(((F32(1) - fx).astype(F32) * F32(2048)).astype(F32)).astype(np.int64)[None, :, None]
        a1 = np.rint((fx * F32(2048)).astype(F32)).astype(np.int64)[None, :, None]
        b0 = np.rint(((F32(1) - fy).astype(F32) * F32(2048)).astype(F32)).astype(np.int64)[:, None, None]
        b1 = np.rint((fy * F32(2048)).astype(F32)).astype(np.int64)[:, None, None]
        s = img.astype(np.int64)
        d0 = s[y0][:, x0] * a0 + s[y0][:, x1] * a1
        d1 = s[y1][:, x0] * a0 + s[y1][:, x1] * a1
        out = np.clip((((b0 * (d0 >> 4)) >> 16) + ((b1 * (d1 >> 4)) >> 16) + 2) >> 2, 0, 255).astype(np.uint8)
    else:
        a0 = (F32(1) - fx).astype(F32)[None, :, None]; a1 = fx[None, :, None]
        b0 = (F32(1) - fy).astype(F32)[:, None, None]; b1 = fy[:, None, None]
        d0 = ((img[y0][:, x0] * a0).astype(F32) + (img[y0][:, x1] * a1).astype(F32)).astype(F32)
        d1 = ((img[y1][:, x0] * a0).astype(F32) + (img[y1][:, x1] * a1).astype(F32)).astype(F32)
        out = ((d0 * b0).astype(F32) + (d1 * b1).astype(F32)).astype(F32)
    return out[:, :, 0] if squeeze else out


def _cv_round(v: np.ndarray) -> np.ndarray:
    """``cvRound(float)`` as OpenCV's x86 builds compute it (``cvtss2si`` / ``cvtps2dq`` in ``v_round``): round half to
    even; a NaN or a value that does not fit a 32-bit integer gives INT_MIN ("integer indefinite") -- after ``>> 5``,
    ``saturate_cast<short>`` and the replicate border that is pixel 0 with a zero fraction, for NaN, +-Inf and huge
    coordinates of either sign."""
    v = np.asarray(v, dtype=F32)
    ok = (v >= F32(-2147483648.0)) & (v < F32(2147483648.0))          # False for NaN
    with np.errstate(invalid="ignore"):
        r = np.rint(np.where(ok, v, F32(0)).astype(F64)).astype(np.int64)
    return np.where(ok, r, np.int64(-2147483648))


def _remap_cv2_compat(src: np.ndarray, map_x: np.ndarray, map_y: np.ndarray) -> np.ndarray:
    """OpenCV ``remap`` INTER_LINEAR restated from its published algorithm
    (modules/imgproc/src/imgwarp.cpp: INTER_BITS=5, INTER_TAB_SIZE=32,
    INTER_REMAP_COEF_BITS=15).  Unverifiable here (no OpenCV)."""
    H, W, C = src.shape
    sx = _cv_round(map_x * F32(32))
    sy = _cv_round(map_y * F32(32))
    ix, fx = sx >> 5, sx & 31
    iy, fy = sy >> 5, sy & 31
    x0 = np.clip(ix, 0, W - 1); x1 = np.clip(ix + 1, 0, W - 1)
    y0 = np.clip(iy, 0, H - 1); y1 = np.clip(iy + 1, 0, H - 1)
    t = (np.arange(32, dtype=F32) * F32(1.0 / 32)).astype(F32)
    one_m = (F32(1) - t).astype(F32)
    # float table tab[fy][fx] = (wy0*wx0, wy0*wx1, wy1*wx0, wy1*wx1)
    w00 = (one_m[:, None] * one_m[None, :]).astype(F32)
    w01 = (one_m[:, None] * t[None, :]).astype(F32)
    w10 = (t[:, None] * one_m[None, :]).astype(F32)
    w11 = (t[:, None] * t[None, :]).astype(F32)
    p00 = src[y0][:, x0]; p01 = src[y0][:, x1]
    p10 = src[y1][:, x0]; p11 = src[y1][:, x1]
    FY, FX = fy[:, None], fx[None, :]
    if src.dtype == np.uint8:
        scale = 1 << 15
        tabs = np.stack([w00, w01, w10, w11], axis=-1).astype(F64) * scale     # exact integers a*b*32
        # saturate_cast<short>: only the (0,0) entry (w00 = 1.0 -> 32768) saturates to 32767.
        # OpenCV then patches that kernel to sum to 2**15 by adding 1 to a zero tap; with integer
        # coordinates the result (p*32767 + q + 2**14) >> 15 equals p either way, so the patch is
        # not observable and is not modelled.
        it = np.clip(np.rint(tabs).astype(np.int64), -32768, 32767)
        acc = (p00.astype(np.int64) * it[FY, FX, 0][..., None] + p01.astype(np.int64) * it[FY, FX, 1][..., None]
               + p10.astype(np.int64) * it[FY, FX, 2][..., None] + p11.astype(np.int64) * it[FY, FX, 3][..., None])
        return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)
    s00 = (p00.astype(F32) * w00[FY, FX][..., None]).astype(F32)
    s01 = (p01.astype(F32) * w01[FY, FX][..., None]).astype(F32)
    s10 = (p10.astype(F32) * w10[FY, FX][..., None]).astype(F32)
    s11 = (p11.astype(F32) * w11[FY, FX][..., None]).astype(F32)
    return (((s00 + s01).astype(F32) + s10).astype(F32) + s11).astype(F32)


# ---------------------------------------------------------------------------
# Composite entry points (the reference's public functions)
# ---------------------------------------------------------------------------

def warp_from_cdf(img: np.ndarray, Fx: np.ndarray, Fy: np.ndarray,
                  out_size: Optional[Tuple[int, int]] = None, mode: str = "cv2") -> np.ndarray:
    """``warp_from_cdf_torch`` (MN/checkpoint_utils.py:133-204) on numpy
    arrays: img [B,C,H,W] (uint8 or float32) -> [B,C,H_out,W_out]."""
    img = np.asarray(img)
    assert img.ndim == 4, f"img must be (B,C,H,W); got {img.shape}"
    B, C, H, W = img.shape
    if Fx.shape[1] != W:
        raise ValueError(f"Fx_img[0] length {Fx.shape[1]} != image width W={W}")
    if Fy.shape[1] != H:
        raise ValueError(f"Fy_img[0] length {Fy.shape[1]} != image height H={H}")
    mx, my = maps_from_cdf(Fx, Fy, out_size)
    outs = [remap_bilinear(np.ascontiguousarray(img[b].transpose(1, 2, 0)), mx[b], my[b], mode) for b in range(B)]
    return np.stack(outs).transpose(0, 3, 1, 2)


def warp_image_by_attention(image: np.ndarray, att_map: np.ndarray, new_width: int, new_height: int,
                            transform: str = "identity", exp_scale: float = 1.0, exp_divisor: float = 1.0,
                            apply_inverse: bool = False, mode: str = "cv2") -> np.ndarray:
    """``warp_image_by_attention`` (AGW/new_method.py:198-283) with the
    module-global transform state passed explicitly."""
    mx, my = maps_from_attention(att_map, new_width, new_height, transform, exp_scale, exp_divisor, apply_inverse)
    return remap_bilinear(image, mx, my, mode)
