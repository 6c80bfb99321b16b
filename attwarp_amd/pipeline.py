"""Batched, device-resident launcher of the hot path.

The reference runs the path one image at a time on the host (``main_batched.py:243-303`` inner loop,
``checkpoint_utils.py:157`` per-sample loop, PNG files in between).  Here a whole batch stays in HBM:

``warp_from_attention_stack``  (BASELINE configs 2-4)
    captured last-query attention rows [T,B,heads,kv]
      -> A1+A2 aggregation                    (attwarp_attn_reduce_stack)
      -> A6 marginals of the 24x24 map        (attwarp_gt_marginals)
      -> A8+A9+A11 PDF -> CDF -> inverse maps (attwarp_axis_maps_from_pdf, one launch, both axes)
      -> A12 bilinear resample                (attwarp_remap_bilinear)

``warp_from_pdf``  (MarginalNet inference chain, MN/trainer.py:285-289)
``warp_from_masks``  (main_batched chain: revise_mask -> uint8 -> LANCZOS -> float64 marginals -> warp)

All launches go to the caller's current stream; no host synchronisation, so the whole step can be
captured in a HIP graph (``capture_step``).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib, _tables
from ._lib import call, ptr, require_gpu, stream_ptr
from . import attention_extraction as ae
from . import checkpoint_utils as cu
from . import new_method as nm

GRID = 24


def adaptive_avg_pool2d(A: torch.Tensor, output_size=(GRID, GRID), sanitize: bool = False) -> torch.Tensor:
    """``F.adaptive_avg_pool2d(A, output_size)`` as the reference calls it on full-resolution attention maps
    (MN/trainer.py:197,433,465): A [B,C,H,W] or [B,H,W] float -> same rank with (H,W) replaced by ``output_size``.
    Window rows ``[floor(i*H/oh), ceil((i+1)*H/oh))``, float64 accumulation rounded once (DESIGN.md "parity").
    ``sanitize=True`` also applies the trainer's ``nan_to_num(nan=0, posinf=0, neginf=0).clamp_min(0)`` (:202)."""
    dev = require_gpu(A)
    if A.dim() not in (3, 4):
        raise ValueError(f"adaptive_avg_pool2d expects [B,C,H,W] or [B,H,W]; got {tuple(A.shape)}")
    oh, ow = (int(output_size), int(output_size)) if isinstance(output_size, int) else (int(output_size[0]), int(output_size[1]))
    x = A.detach().float().contiguous()
    H, W = x.shape[-2], x.shape[-1]
    planes = x.numel() // (H * W)
    out = torch.empty(*x.shape[:-2], oh, ow, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        call("attwarp_adaptive_avg_pool", ptr(x), planes, H, W, oh, ow, int(bool(sanitize)), ptr(out), stream_ptr(dev))
    return out


def axis_maps_from_pdf(px: torch.Tensor, py: torch.Tensor, size_hw: Tuple[int, int],
                       out_size: Optional[Tuple[int, int]] = None, eps: float = 1e-8
                       ) -> Tuple[torch.Tensor, torch.Tensor]:
    """px [B,n], py [B,n] low-resolution PDFs -> (map_x [B,W_out], map_y [B,H_out]).
    Equals upsample_pdf_right_inverse(.).clamp_min(0) -> cdf_from_density -> grid construction."""
    dev = require_gpu(px, py)
    H, W = int(size_hw[0]), int(size_hw[1])
    H_out, W_out = (H, W) if out_size is None else (int(out_size[0]), int(out_size[1]))
    x, y = px.detach().float().contiguous(), py.detach().float().contiguous()
    B, n = x.shape
    inv_x = _tables.right_inverse_inv(n, W, eps, dev)
    inv_y = _tables.right_inverse_inv(n, H, eps, dev)
    mx = torch.empty(B, W_out, device=dev, dtype=torch.float32)
    my = torch.empty(B, H_out, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        call("attwarp_axis_maps_from_pdf", ptr(x), ptr(y), B, n, W, H, W_out, H_out, ptr(inv_x), ptr(inv_y), ptr(mx),
             ptr(my), stream_ptr(dev))
    return mx, my


def warp_from_pdf(images: torch.Tensor, px: torch.Tensor, py: torch.Tensor, out_size=None, channels_last=False,
                  mode: str = "cv2") -> torch.Tensor:
    """images [B,C,H,W] (or [B,H,W,C]) float32/uint8; px [B,24] over x, py [B,24] over y."""
    if channels_last:
        H, W = images.shape[1], images.shape[2]
    else:
        H, W = images.shape[2], images.shape[3]
    mx, my = axis_maps_from_pdf(px, py, (H, W), out_size)
    return cu.remap_separable(images, mx, my, mode=mode, channels_last=channels_last)


def axis_maps_from_attention_steps(steps: torch.Tensor, size_hw: Tuple[int, int],
                                   out_size: Optional[Tuple[int, int]] = None, eps: float = 1e-8,
                                   return_attention: bool = False, maps_out=None):
    """Per-step aggregated maps [T,B,g*g] in the attention dtype (output of the A1 kernel: float32 / float16 /
    bfloat16) -> (map_x, map_y) in ONE launch: mean over steps (rounded in that dtype, llava.py:409-411), marginals of the
    g x g map, PDF up-sample, CDF, inverse maps.  Bit-identical to attn_finalize -> float() -> gt_marginals ->
    axis_maps_from_pdf."""
    dev = require_gpu(steps)
    s = steps.detach().contiguous()
    if s.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        raise TypeError("axis_maps_from_attention_steps: float32 / float16 / bfloat16 step maps expected")
    T, B, ntok = s.shape
    g = int(round(ntok ** 0.5))
    if g * g != ntok:
        raise ValueError(f"step maps of {ntok} tokens are not a square grid")
    H, W = int(size_hw[0]), int(size_hw[1])
    H_out, W_out = (H, W) if out_size is None else (int(out_size[0]), int(out_size[1]))
    inv_x = _tables.right_inverse_inv(g, W, eps, dev)
    inv_y = _tables.right_inverse_inv(g, H, eps, dev)
    if maps_out is None:
        mx = torch.empty(B, W_out, device=dev, dtype=torch.float32)
        my = torch.empty(B, H_out, device=dev, dtype=torch.float32)
    else:                       # caller-owned static buffers (graph replay)
        mx, my = maps_out
        if tuple(mx.shape) != (B, W_out) or tuple(my.shape) != (B, H_out) or mx.dtype != torch.float32 or my.dtype != torch.float32:
            raise ValueError("axis_maps_from_attention_steps: maps_out must be float32 [B,W_out], [B,H_out]")
        if not (mx.is_contiguous() and my.is_contiguous()) or mx.device != dev or my.device != dev:
            raise ValueError("axis_maps_from_attention_steps: maps_out must be dense tensors on the device of `steps` "
                             "(the kernel writes dense [B,W_out] / [B,H_out] rows through the raw pointers)")
    att = torch.empty(B, ntok, device=dev, dtype=torch.float32) if return_attention else None
    with torch.cuda.device(dev):
        call("attwarp_axis_maps_from_steps_t", ptr(s), _lib.dtype_id(s), T, B, g, W, H, W_out, H_out, ptr(inv_x), ptr(inv_y),
             ptr(mx), ptr(my), ptr(att), stream_ptr(dev))
    return (mx, my, att) if return_attention else (mx, my)


def attention_step_maps(rows: torch.Tensor, starts: torch.Tensor, ntok: int = ae.NUM_IMAGE_TOKENS,
                        starts_tiled: Optional[torch.Tensor] = None) -> torch.Tensor:
    """A1 over a captured stack: rows [T,B,heads,kv] -> per-step maps [T,B,ntok] (one launch)."""
    r = rows.contiguous()
    T, B, heads, kv = r.shape
    if starts_tiled is None:
        starts_tiled = starts.repeat(T)
    steps = ae.attn_reduce_step(r.view(T * B, heads, 1, kv), starts_tiled, ntok)
    return steps.view(T, B, ntok)


def warp_from_attention_stack(images: torch.Tensor, rows: torch.Tensor, starts: torch.Tensor, out_size=None,
                              channels_last=False, mode: str = "cv2", out: Optional[torch.Tensor] = None,
                              starts_tiled: Optional[torch.Tensor] = None) -> torch.Tensor:
    """images: batch on the GPU; rows [T,B,heads,kv] last-query attention rows; starts int32 [B].
    Three launches: A1 (step maps) -> fused A2+A6+A8+A9+A11 (maps) -> A12 (warp), for attention in any of the three dtypes."""
    if channels_last:
        H, W = images.shape[1], images.shape[2]
    else:
        H, W = images.shape[2], images.shape[3]
    # any attention dtype (float32 / the model's float16 / bfloat16): A1 per step, then the fused A2 .. A11 launch
    steps = attention_step_maps(rows, starts, ae.NUM_IMAGE_TOKENS, starts_tiled)
    mx, my = axis_maps_from_attention_steps(steps, (H, W), out_size)
    return cu.remap_separable(images, mx, my, mode=mode, channels_last=channels_last, out=out)


def warp_from_masks(images_u8: torch.Tensor, attn24: torch.Tensor, out_size=(500, 500), enhance_coe=10,
                    kernel_size=3, transform="identity", exp_scale=1.0, exp_divisor=1.0, apply_inverse=False,
                    mode: str = "cv2") -> torch.Tensor:
    """The ``main_batched.py:243-287`` chain for a batch of equally sized images.
    images_u8 [B,H,W,3] uint8 (channel order is irrelevant to the warp); attn24 [B,24,24].
    -> [B,H_out,W_out,3] uint8."""
    B, H, W, _ = images_u8.shape
    rev = ae.revise_mask(attn24.float(), kernel_size=kernel_size, enhance_coe=enhance_coe)
    mota = ae.upsample_mask_lanczos(rev, (W, H))                       # uint8 [B,H,W]
    mx, my = nm.attention_axis_maps(mota, out_size[1], out_size[0], transform, exp_scale, exp_divisor, apply_inverse)
    return nm.remap_hwc(images_u8, mx, my, mode)


@torch.no_grad()
def warp_from_marginalnet(net, fmap_v: torch.Tensor, txt_tok: torch.Tensor, txt_mask: torch.Tensor,
                          images: torch.Tensor, out_size=None, channels_last=False, mode: str = "cv2"):
    """BASELINE config 5's device-resident chain: MarginalNet(hidden=256) forward (stock PyTorch-ROCm ops +
    the HIP safe_softmax) -> px, py over the 24 x 24 grid -> A8+A9+A11 (one launch) -> A12 warp.
    Mirrors the inference block of the reference trainer (MN/trainer.py:210, :285-289).
    Returns (warped images, px, py)."""
    px, py = net(fmap_v, GRID, GRID, txt_tok, txt_mask)
    return warp_from_pdf(images, px, py, out_size, channels_last, mode), px, py


OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def expand2square(images_u8: torch.Tensor, background=None) -> torch.Tensor:
    """LLaVA-1.5's ``image_aspect_ratio == "pad"`` step (``expand2square`` of the external LLaVA checkout, called by
    ``process_images``: AGW/attention_extraction/functions.py:262-271) for a uint8 [B,H,W,C] batch on the GPU: the
    images centred (offset ``(long - short) // 2``) on a square canvas of ``background`` (default
    ``int(mean * 255)`` per channel, as LLaVA passes it).  Square batches are returned as they are."""
    dev = require_gpu(images_u8)
    if images_u8.dtype != torch.uint8 or images_u8.dim() != 4:
        raise TypeError("expand2square expects a uint8 [B,H,W,C] tensor")
    B, H, W, C = images_u8.shape
    if H == W:
        return images_u8
    if background is None:
        background = tuple(int(x * 255) for x in OPENAI_CLIP_MEAN)
    if len(background) < C:
        raise ValueError(f"background needs {C} channel values; got {len(background)}")
    n = max(H, W)
    out = torch.tensor(list(background[:C]), dtype=torch.uint8, device=dev).expand(B, n, n, C).contiguous()
    if W > H:
        o = (W - H) // 2
        out[:, o:o + H] = images_u8
    else:
        o = (H - W) // 2
        out[:, :, o:o + W] = images_u8
    return out


def clip_preprocess(images_u8: torch.Tensor, size: int = 336, dtype: torch.dtype = torch.float16,
                    mean=OPENAI_CLIP_MEAN, std=OPENAI_CLIP_STD, pad_to_square: bool = False) -> torch.Tensor:
    """Warped uint8 RGB batch [B,H,W,3] -> CLIP-ready [B,3,size,size] on the GPU ("next" row 3).

    Replaces the PNG round trip of the reference (``cv2.imwrite`` new_method.py:491 -> ``Image.open`` +
    ``process_images`` evaluate_accuracy.py:157-158) with two launches.  Arithmetic = HF CLIPImageProcessor
    (PIL backend) as LLaVA-1.5 configures it: PIL BICUBIC resize of the shorter edge to ``size``, center crop,
    ``float32(float64(u8) * (1/255))``, ``(x - mean) / std`` in float32; ``dtype`` float32 or float16
    (LLaVA casts to float16, functions.py:270-271).  Bit-identical to the processor for float32.
    ``pad_to_square=True`` runs :func:`expand2square` first (LLaVA-1.5's ``image_aspect_ratio="pad"``; a no-op for
    the square warps of the reference drivers)."""
    import ctypes
    dev = require_gpu(images_u8)
    if images_u8.dtype != torch.uint8 or images_u8.dim() != 4:
        raise TypeError("clip_preprocess expects a uint8 [B,H,W,C] tensor")
    if pad_to_square:
        images_u8 = expand2square(images_u8, tuple(int(float(v) * 255) for v in mean))
    if dtype not in (torch.float32, torch.float16):
        raise TypeError("clip_preprocess: dtype must be float32 or float16")
    x = images_u8.contiguous()
    B, H, W, C = x.shape
    short, long_ = (H, W) if H <= W else (W, H)
    new_long = int(size * long_ / short)
    nh, nw = (size, new_long) if H <= W else (new_long, size)
    top, left = (nh - size) // 2, (nw - size) // 2
    bx, kx, ksx = _tables.lanczos_tables(W, nw, dev, "bicubic")
    by, ky, ksy = _tables.lanczos_tables(H, nh, dev, "bicubic")
    tmp = torch.empty(B, H, size, C, device=dev, dtype=torch.uint8)
    out = torch.empty(B, C, size, size, device=dev, dtype=dtype)
    m = (ctypes.c_float * C)(*[float(v) for v in mean[:C]])
    s = (ctypes.c_float * C)(*[float(v) for v in std[:C]])
    with torch.cuda.device(dev):
        call("attwarp_clip_preprocess_u8", ptr(x), B, H, W, C, top, left, size, ptr(bx), ptr(kx), ksx, ptr(by), ptr(ky),
             ksy, ctypes.cast(m, ctypes.c_void_p), ctypes.cast(s, ctypes.c_void_p), ptr(tmp), ptr(out),
             _lib.F32 if dtype == torch.float32 else _lib.F16, stream_ptr(dev))
    return out


def random_clip_vision_tower(device, dtype: torch.dtype = torch.float16, seed: int = 0):
    """The architecture of LLaVA-1.5's vision tower (openai/clip-vit-large-patch14-336: ViT-L/14 at 336 x 336, 24 layers,
    hidden 1024, 16 heads, MLP 4096) from ``transformers`` with SEEDED RANDOM weights -- the checkpoint, the ``llava``
    package and TextVQA are absent here (no network), so BASELINE configs[4] can be closed as a data flow with the right
    shapes and dtypes, not as an accuracy number.  Stock PyTorch-ROCm ops (MFMA-class library GEMMs): outside the
    hand-kernel scope, like MarginalNet's convolutions (SURVEY 8d "bounding roofline")."""
    from transformers import CLIPVisionConfig, CLIPVisionModel
    cfg = CLIPVisionConfig(image_size=336, patch_size=14, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                           intermediate_size=4096, projection_dim=768)
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        tower = CLIPVisionModel(cfg)
    finally:
        torch.random.set_rng_state(gen_state)
    return tower.to(device=device, dtype=dtype).eval()


@torch.no_grad()
def vision_tower_features(tower, pixel_values: torch.Tensor, select_layer: int = -2) -> torch.Tensor:
    """LLaVA-1.5's feature selection (``mm_vision_select_layer = -2``, ``mm_vision_select_feature = "patch"``; the reference
    feeds the warped image through it at AGW/evaluate_accuracy.py:157-178): pixel_values [B,3,336,336] ->
    ``hidden_states[-2][:, 1:]`` [B,576,1024] -> the token map [B,1024,24,24] MarginalNet consumes (MN/model.py:55-68)."""
    out = tower(pixel_values=pixel_values.to(next(tower.parameters()).dtype), output_hidden_states=True)
    tok = out.hidden_states[select_layer][:, 1:]
    B, n, d = tok.shape
    g = int(round(n ** 0.5))
    return tok.transpose(1, 2).reshape(B, d, g, g)


@torch.no_grad()
def config5_chain(tower, net, images_u8: torch.Tensor, txt_tok: torch.Tensor, txt_mask: torch.Tensor, out_size=(500, 500),
                  mode: str = "cv2", record=None):
    """BASELINE configs[4] as a device-resident data flow (nothing leaves the GPU, no PNG round trip):

        images_u8 [B,H,W,3] -> CLIP tensor -> vision tower -> token map [B,1024,24,24]                  (the model's own pass)
          -> MarginalNet(1024, 4096, 256)(token map, text tokens) -> px, py [B,24]                     MN/model.py:55-95
          -> A8 + A9 + A11 -> A12: warp of the uint8 images to ``out_size``                             MN/trainer.py:285-289
          -> CLIP tensor of the WARPED images [B,3,336,336] -> vision tower -> features [B,1024,24,24]  evaluate_accuracy.py:157-178

    ``record``: a callable ``record(name)`` invoked after each leg (bench.py records HIP events there).  Returns a dict of
    every intermediate.  Parity of each hand-written leg is pinned elsewhere (clip_preprocess, MarginalNet tail, the warp);
    TextVQA accuracy parity is unobtainable here: LLaVA-1.5-7B's weights, the ``llava`` package and the dataset are absent."""
    rec = record or (lambda name: None)
    dtype = next(tower.parameters()).dtype
    pix0 = clip_preprocess(images_u8, 336, dtype, pad_to_square=True)
    rec("clip_tensor_in")
    fmap = vision_tower_features(tower, pix0)
    rec("tower_in")
    px, py = net(fmap, GRID, GRID, txt_tok, txt_mask)
    rec("marginalnet")
    B, H, W, _ = images_u8.shape
    mx, my = axis_maps_from_pdf(px, py, (H, W), out_size)
    warped = cu.remap_separable(images_u8, mx, my, mode=mode, channels_last=True)
    rec("warp")
    pix1 = clip_preprocess(warped, 336, dtype, pad_to_square=True)
    rec("clip_tensor_warped")
    feats = vision_tower_features(tower, pix1)
    rec("tower_warped")
    return {"pixel_values_in": pix0, "token_map": fmap, "px": px, "py": py, "map_x": mx, "map_y": my, "warped": warped,
            "pixel_values_warped": pix1, "features_warped": feats}


def capture_step(fn, *args, warmup: int = 2):
    """Capture ``fn(*args)`` (a function made only of this package's launches on static tensors) into a
    HIP graph; returns (graph, output).  ``graph.replay()`` re-runs the step with one host call."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(warmup):
            out = fn(*args)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn(*args)
    return g, out


class OverlappedWarp:
    """Steady-state form of ``warp_from_attention_stack`` for a stream of equally shaped batches.  The three kernels of
    a step belong to three different batches and are independent --
        R: the resample of batch k                      (HBM-bound, maps[c])
        M: the map construction of batch k+1            (latency-bound, steps[c] -> maps[1-c])
        A: the attention reduce of batch k+2            (a second bandwidth stream, rows -> steps[1-c])
    -- so they need not run reduce -> maps -> resample back to back for every batch.  ``pattern``:
        "fused": R | M | A as block ranges of ONE launch (attwarp_warp_step_fused; float32 images on the staged resample)
        "am":    A + M as one launch (attwarp_attn_reduce_and_maps), then R on its own (images of >= 4 MB, where the
                 resample is > 90 % of the step and runs best with its stand-alone dispatch; any image dtype)
        "dag":   R and A on one stream, M on a side stream with exactly the edges the buffers need
        "join":  three graph branches forked and joined in every step
        "auto":  "am" for float32 images of >= 4 MB, else "fused" when eligible (float32 images), else "am".
    Steps are replayed as HIP graphs over static buffers; the per-step maps and attention maps are double buffered.

    ``images`` / ``rows`` are either one static buffer each (the caller copies every batch in) or RINGS of n buffers
    (lists of equal length): batch k lives in ``images[k % n]`` / ``rows[k % n]`` and is warped into ``outs[k % n]``,
    so a producer can fill slot (k+1) % n while slot k % n is being read, and a measurement over a ring larger than the
    Infinity Cache streams every batch from HBM.

        ow = OverlappedWarp(images, rows, starts)      # allocates outs, captures lazily
        rows[0] <- attention of batch 0;  ow.prime()   # reduce + maps of batch 0 (serial)
        rows[1 % n] <- attention of batch 1;  ow.prime2()   # reduce of batch 1 (serial)
        for k in 0 .. N-1:
            images[k % n] <- batch k;  rows[(k+2) % n] <- attention of batch k+2 (if any)
            out = ow.step()                            # R(k) || M(k+1) || A(k+2)   -> outs[k % n]
    i.e. the attention runs two batches ahead of the images (``ow.flush()`` = R alone, for a tail).  Every step is
    bit-identical to ``warp_from_attention_stack`` on its batch (same kernels, same arguments)."""

    def __init__(self, images, rows, starts: torch.Tensor, out_size=None, channels_last: bool = False,
                 mode: str = "cv2", pattern: str = "auto"):
        if pattern not in ("auto", "fused", "am", "dag", "join"):
            raise ValueError("OverlappedWarp: pattern must be 'auto', 'fused', 'am', 'dag' or 'join'")
        self.pattern = pattern
        self.images = list(images) if isinstance(images, (list, tuple)) else [images]
        self.rows = list(rows) if isinstance(rows, (list, tuple)) else [rows]
        if len(self.images) != len(self.rows) or not self.images:
            raise ValueError("OverlappedWarp: images and rows must be rings of the same length")
        dev = require_gpu(*self.images, *self.rows, starts)
        if not all(t.is_contiguous() for t in self.images + self.rows + [starts]):
            raise ValueError("OverlappedWarp: static input buffers must be contiguous (the graph holds raw pointers)")
        if any(r.dtype not in (torch.float32, torch.float16, torch.bfloat16) or r.dtype != self.rows[0].dtype or r.dim() != 4
               or r.shape != self.rows[0].shape for r in self.rows):
            raise TypeError("OverlappedWarp: attention rows [T,B,heads,kv] of one shape and one dtype (float32 / float16 / "
                            "bfloat16) expected")
        if any(i.shape != self.images[0].shape or i.dtype != self.images[0].dtype for i in self.images):
            raise ValueError("OverlappedWarp: every image buffer of the ring must have the same shape and dtype")
        self.n = len(self.images)
        self.starts = starts.clone()         # own copy: set_starts() writes into it
        self.channels_last, self.mode = channels_last, mode
        img0 = self.images[0]
        H, W = (img0.shape[1], img0.shape[2]) if channels_last else (img0.shape[2], img0.shape[3])
        self.size_hw = (H, W)
        self.out_size = out_size
        T = self.rows[0].shape[0]
        self.starts_tiled = starts.repeat(T).contiguous()
        B = img0.shape[0]
        Ho, Wo = (H, W) if out_size is None else out_size
        shape = (B, Ho, Wo, img0.shape[3]) if channels_last else (B, img0.shape[1], Ho, Wo)
        self.outs = [torch.empty(shape, device=dev, dtype=img0.dtype) for _ in range(self.n)]
        self.cur = 0                 # which (steps, maps) buffer set the next resample reads
        self.k = 0                   # ring position of the next resample
        self._dev = dev
        self._sideM = torch.cuda.Stream(device=dev)
        self._sideA = torch.cuda.Stream(device=dev)
        self._graphs = {}
        # both buffer sets, filled once outside capture (allocations, lazily built tables)
        self.steps = [attention_step_maps(self.rows[0], self.starts, ae.NUM_IMAGE_TOKENS, self.starts_tiled) for _ in (0, 1)]
        self.maps = [axis_maps_from_attention_steps(self.steps[i], self.size_hw, self.out_size) for i in (0, 1)]
        cu.remap_separable(self.images[0], *self.maps[0], mode=self.mode, channels_last=self.channels_last, out=self.outs[0])
        torch.cuda.synchronize(dev)
        g = int(round(ae.NUM_IMAGE_TOKENS ** 0.5))
        self._inv = (_tables.right_inverse_inv(g, W, 1e-8, dev), _tables.right_inverse_inv(g, H, 1e-8, dev))
        if self.pattern == "auto" and img0.dtype == torch.float32 and img0[0].numel() * 4 >= (4 << 20):
            # large images (>= 4 MB each): the resample is > 90 % of a step and keeps its own launch (inside the fused launch it runs 2 %
            # slower); the reduce and the latency-bound map construction share one (attwarp_attn_reduce_and_maps)
            self.pattern = "am"
        if self.pattern in ("auto", "fused"):
            # one launch per step (attwarp_warp_step_fused) when the shapes are eligible: float32 images on the staged
            # resample, attention rows in any of the three dtypes
            try:
                if img0.dtype != torch.float32:
                    raise _lib.AttWarpError("fused step: float32 images only")
                self._fused_step(0, 0)
                torch.cuda.synchronize(dev)
                self.pattern = "fused"
            except _lib.AttWarpError:
                if self.pattern == "fused":
                    raise
                # measured (tools/attic/pattern_probe.py, uint8 images): "am" 0.040 / 0.124 ms per step at B=64 / 256 336x336
                # against 0.048 / 0.131 ("dag") and 0.048 / 0.125 ("join")
                self.pattern = "am"
            # the trial step overwrote maps[1] / steps[1] with what they held anyway (same inputs)

    def _fused_step(self, c, k):
        """R(k) | M(k+1) | A(k+2) as ONE launch (block ranges of one grid)."""
        img, out = self.images[k % self.n], self.outs[k % self.n]
        rows = self.rows[(k + 2) % self.n]
        T, B, heads, kv = rows.shape
        if self.channels_last:
            _, H, W, C = img.shape
            _, Ho, Wo, _ = out.shape
        else:
            _, C, H, W = img.shape
            _, _, Ho, Wo = out.shape
        mx, my = self.maps[c]
        nx, ny = self.maps[1 - c]
        g = int(round(ae.NUM_IMAGE_TOKENS ** 0.5))
        with torch.cuda.device(self._dev):
            call("attwarp_warp_step_fused", ptr(img), ptr(out), _lib.HWC if self.channels_last else _lib.CHW, B, C, H, W,
                 Ho, Wo, ptr(mx), ptr(my), _lib.MODE_IDS[self.mode],
                 _lib.dtype_id(rows), ptr(self.steps[c]), T, g, ptr(self._inv[0]), ptr(self._inv[1]), ptr(nx), ptr(ny),
                 ptr(rows), T * B, heads, kv, ptr(self.starts_tiled), T * B, ae.NUM_IMAGE_TOKENS, ptr(self.steps[1 - c]),
                 stream_ptr(self._dev))

    @property
    def out(self) -> torch.Tensor:
        """Output buffer of the most recent step."""
        return self.outs[(self.k - 1) % self.n]

    def _am_step(self, c, k):
        """A(k+2) + M(k+1) as one launch, then R(k)."""
        self._am_launch(c, k)
        self._resample(c, k % self.n)

    def _am_launch(self, c, k):
        """A(k+2): rows[k+2] -> steps[1-c]  and  M(k+1): steps[c] -> maps[1-c]  as ONE launch."""
        rows = self.rows[(k + 2) % self.n]
        T, B, heads, kv = rows.shape
        H, W = self.size_hw
        Ho, Wo = (H, W) if self.out_size is None else self.out_size
        g = int(round(ae.NUM_IMAGE_TOKENS ** 0.5))
        nx, ny = self.maps[1 - c]
        with torch.cuda.device(self._dev):
            call("attwarp_attn_reduce_and_maps", _lib.dtype_id(rows), ptr(rows), T * B, heads, kv, ptr(self.starts_tiled), T * B,
                 ae.NUM_IMAGE_TOKENS, ptr(self.steps[1 - c]), ptr(self.steps[c]), T, B, g, W, H, Wo, Ho, ptr(self._inv[0]),
                 ptr(self._inv[1]), ptr(nx), ptr(ny), stream_ptr(self._dev))

    def _reduce(self, i, r):
        """A: rows[r] -> steps[i]"""
        T, B, heads, kv = self.rows[r].shape
        ae.attn_reduce_step(self.rows[r].view(T * B, heads, 1, kv), self.starts_tiled, ae.NUM_IMAGE_TOKENS,
                            out=self.steps[i].view(T * B, ae.NUM_IMAGE_TOKENS))

    def _maps(self, i_steps, i_maps):
        """M: steps[i_steps] -> maps[i_maps]"""
        axis_maps_from_attention_steps(self.steps[i_steps], self.size_hw, self.out_size, maps_out=self.maps[i_maps])

    def _resample(self, i, r):
        cu.remap_separable(self.images[r], *self.maps[i], mode=self.mode, channels_last=self.channels_last,
                           out=self.outs[r])

    def _capture(self, c, k, unroll: int = 1):
        """graph: ``unroll`` consecutive steps starting with buffer set c at ring position k.  Step u launches
        R(k): resample images[k] with maps[c] | M(k+1): steps[c] -> maps[1-c] | A(k+2): rows[k+2] -> steps[1-c].
        pattern "dag": R and A alternate on ONE stream (both are bandwidth-bound: side by side they would only share
        the same HBM), the latency-bound M runs on a side stream with exactly the edges the buffers need -- M(k+1) after
        A(k+1) and R(k-1) (which read maps[1-c]), R(k+1) after M(k+1), A(k+3) after M(k+1) (which read steps[c]) -- so the
        main stream never waits for a kernel that has not long finished.  pattern "join": three branches forked and
        joined in every step."""
        g = torch.cuda.CUDAGraph()
        main = torch.cuda.Stream(device=self._dev)
        main.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(main):
            with torch.cuda.graph(g, stream=main):
                if self.pattern in ("fused", "am"):
                    for u in range(unroll):
                        (self._fused_step if self.pattern == "fused" else self._am_step)(c, k)
                        c ^= 1
                        k += 1
                elif self.pattern == "dag":
                    for u in range(unroll):
                        # main so far: ..., A(k+1), R(k-1) [previous step: R then A]; side: ..., M(k)
                        self._sideM.wait_stream(main)                # M(k+1) after A(k+1) and R(k-1)
                        with torch.cuda.stream(self._sideM):
                            self._maps(c, 1 - c)
                        self._resample(c, k % self.n)                # R(k) needs M(k): joined at the end of step u-1
                        self._reduce(1 - c, (k + 2) % self.n)        # A(k+2) overwrites steps[1-c], read by M(k): done
                        main.wait_stream(self._sideM)                # next R needs M(k+1); next A overwrites steps[c]
                        c ^= 1
                        k += 1
                else:
                    for u in range(unroll):
                        self._sideM.wait_stream(main)                # fork
                        self._sideA.wait_stream(main)
                        with torch.cuda.stream(self._sideA):
                            self._reduce(1 - c, (k + 2) % self.n)
                        with torch.cuda.stream(self._sideM):
                            self._maps(c, 1 - c)
                        self._resample(c, k % self.n)
                        main.wait_stream(self._sideM)                # join
                        main.wait_stream(self._sideA)
                        c ^= 1
                        k += 1
        torch.cuda.current_stream().wait_stream(main)
        return g

    def _graph(self, unroll):
        key = (self.cur, self.k % self.n, unroll)
        if key not in self._graphs:
            self._graphs[key] = self._capture(self.cur, self.k % self.n, unroll)
        return self._graphs[key]

    def run(self, n: int, unroll: int = 8) -> torch.Tensor:
        """n steps on the buffers as they are (steady-state measurement / a producer that stays ahead of the ring):
        graphs of ``unroll`` steps (one host call per ``unroll`` steps), then single steps for the remainder.  With a
        ring of n buffers a graph of lcm(2, n)-multiple length returns to the state it started from and is re-used."""
        unroll -= unroll % 2
        if unroll >= 2 and self.n > 1:
            import math
            period = math.lcm(2, self.n)
            unroll = max(period, unroll - unroll % period)
        while unroll >= 2 and n >= unroll:
            self._graph(unroll).replay()
            self.cur ^= unroll & 1
            self.k += unroll
            n -= unroll
        for _ in range(n):
            self.step()
        return self.out

    def set_starts(self, starts: torch.Tensor):
        """New first-image-token positions ([B] int32) for the batches reduced from now on: copied into the static
        buffer the captured graphs read (stream ordered; the attention runs two batches ahead of the images, so a
        stream whose prompts change per batch calls this before the step that reduces that batch)."""
        if starts.shape != self.starts.shape or starts.dtype != self.starts.dtype or starts.device != self.starts.device:
            raise ValueError("OverlappedWarp.set_starts: same shape, dtype and device as the starts given at construction")
        self.starts.copy_(starts)
        self.starts_tiled.copy_(starts.repeat(self.rows[0].shape[0]))

    def reset(self):
        """Back to ring position 0 / buffer set 0 (a new stream on the same static buffers; prime() again)."""
        self.cur = 0
        self.k = 0

    def prime(self):
        """Reduce + maps of the batch at the current ring position into the current buffer set (serial, no graph)."""
        self._reduce(self.cur, self.k % self.n)
        self._maps(self.cur, self.cur)

    def prime2(self):
        """Reduce of the batch after it into the current steps buffer (its maps are built by the first step)."""
        self._reduce(self.cur, (self.k + 1) % self.n)

    def step(self) -> torch.Tensor:
        self._graph(1).replay()
        self.cur ^= 1
        self.k += 1
        return self.out

    def flush(self) -> torch.Tensor:
        """Resample of the batch at the current ring position with the current maps alone, then advance (tail of a
        stream: no further attention to reduce)."""
        self._resample(self.cur, self.k % self.n)
        self.k += 1
        return self.out

    def tail(self) -> torch.Tensor:
        """The last two batches of a stream that ends: R(N-2) + M(N-1), then R(N-1)."""
        self._resample(self.cur, self.k % self.n)
        self._maps(self.cur, 1 - self.cur)
        self.cur ^= 1
        self.k += 1
        return self.flush()


class _StepSlot(__import__("ctypes").Structure):
    """``attwarp_step_slot`` of include/attwarp.h (device pointers)."""
    _fields_ = [(n, __import__("ctypes").c_void_p) for n in ("src", "dst", "map_x", "map_y", "steps_in", "map_x_next", "map_y_next",
                                                             "rows", "starts", "steps_out")]


class PairedStepWarp:
    """``OverlappedWarp`` pattern "fused" with TWO batches of the stream per piece and launch
    (attwarp_warp_step_fused_slots):

        launch p:   R(2p), R(2p+1)  |  M(2p+2), M(2p+3)  |  A(2p+4), A(2p+5)

    Every dependency is on an earlier launch, and the launch's ramp and tail -- about 8 us of a 60 us step at B=64
    336x336 -- are paid once per two batches.  Per-step maps and step-map buffers exist four times (batch j uses set
    j % 4); ``images`` / ``rows`` are rings of n >= 2 static buffers (even n), batch k in slot k % n, output in
    ``outs[k % n]``.  float32 images on the staged resample, attention rows float32 / float16 / bfloat16.

        pw = PairedStepWarp(images, rows, starts)
        rows[0..3] <- attention of batches 0..3;  pw.prime()         # reduce of 0..3, maps of 0, 1 (serial launches)
        for p in 0 .. N/2 - 1:
            images[2p % n], images[(2p+1) % n] <- batches 2p, 2p+1;  rows[(2p+4) % n], rows[(2p+5) % n] <- attention ahead
            pw.step()                                                  # -> outs[2p % n], outs[(2p+1) % n]
        pw.tail()                                                      # the last four batches without further attention

    Bit-identical, batch by batch, to ``warp_from_attention_stack`` (same kernels bodies, same arguments)."""

    def __init__(self, images, rows, starts: torch.Tensor, out_size=None, channels_last: bool = False, mode: str = "cv2"):
        self.images, self.rows = list(images), list(rows)
        if len(self.images) != len(self.rows) or len(self.images) < 2 or len(self.images) % 2:
            raise ValueError("PairedStepWarp: images and rows must be rings of the same even length >= 2")
        dev = require_gpu(*self.images, *self.rows, starts)
        img0, r0 = self.images[0], self.rows[0]
        if any(i.dtype != torch.float32 or i.shape != img0.shape or not i.is_contiguous() for i in self.images):
            raise TypeError("PairedStepWarp: contiguous float32 image buffers of one shape expected")
        if any(r.dtype not in (torch.float32, torch.float16, torch.bfloat16) or r.dtype != r0.dtype or r.shape != r0.shape
               or r.dim() != 4 or not r.is_contiguous() for r in self.rows):
            raise TypeError("PairedStepWarp: contiguous attention rows [T,B,heads,kv] of one shape and dtype expected")
        self.n = len(self.images)
        self.channels_last, self.mode, self._dev = channels_last, mode, dev
        H, W = (img0.shape[1], img0.shape[2]) if channels_last else (img0.shape[2], img0.shape[3])
        self.C = img0.shape[3] if channels_last else img0.shape[1]
        self.B, self.H, self.W = img0.shape[0], H, W
        self.Ho, self.Wo = (H, W) if out_size is None else (int(out_size[0]), int(out_size[1]))
        T = r0.shape[0]
        self.starts = starts.clone()
        self.starts_tiled = starts.repeat(T).contiguous()
        shape = (self.B, self.Ho, self.Wo, self.C) if channels_last else (self.B, self.C, self.Ho, self.Wo)
        self.outs = [torch.empty(shape, device=dev, dtype=torch.float32) for _ in range(self.n)]
        self.steps = [torch.empty(T, self.B, ae.NUM_IMAGE_TOKENS, device=dev, dtype=r0.dtype) for _ in range(4)]
        self.maps = [(torch.empty(self.B, self.Wo, device=dev, dtype=torch.float32),
                      torch.empty(self.B, self.Ho, device=dev, dtype=torch.float32)) for _ in range(4)]
        g = int(round(ae.NUM_IMAGE_TOKENS ** 0.5))
        self.g = g
        self._inv = (_tables.right_inverse_inv(g, W, 1e-8, dev), _tables.right_inverse_inv(g, H, 1e-8, dev))
        self.k = 0
        self._graphs = {}
        self._launch(0)                        # eligibility (raises AttWarpError) + lazy module loading, outside any capture
        torch.cuda.synchronize(dev)

    def _launch(self, k):
        """R(k), R(k+1) | M(k+2), M(k+3) | A(k+4), A(k+5)"""
        import ctypes
        T, B, heads, kv = self.rows[0].shape
        slots = (_StepSlot * 2)()
        for s in (0, 1):
            j = k + s
            mx, my = self.maps[j % 4]
            nx, ny = self.maps[(j + 2) % 4]
            slots[s] = _StepSlot(ptr(self.images[j % self.n]), ptr(self.outs[j % self.n]), ptr(mx), ptr(my),
                                 ptr(self.steps[(j + 2) % 4]), ptr(nx), ptr(ny),
                                 ptr(self.rows[(j + 4) % self.n]), ptr(self.starts_tiled), ptr(self.steps[j % 4]))
        with torch.cuda.device(self._dev):
            call("attwarp_warp_step_fused_slots", ctypes.cast(slots, ctypes.c_void_p), 2,
                 _lib.HWC if self.channels_last else _lib.CHW, B, self.C, self.H, self.W, self.Ho, self.Wo, _lib.MODE_IDS[self.mode],
                 _lib.dtype_id(self.rows[0]), T, self.g, ptr(self._inv[0]), ptr(self._inv[1]), T * B, heads, kv, T * B,
                 ae.NUM_IMAGE_TOKENS, stream_ptr(self._dev))

    def _reduce(self, j):
        T, B, heads, kv = self.rows[0].shape
        ae.attn_reduce_step(self.rows[j % self.n].view(T * B, heads, 1, kv), self.starts_tiled, ae.NUM_IMAGE_TOKENS,
                            out=self.steps[j % 4].view(T * B, ae.NUM_IMAGE_TOKENS))

    def _maps(self, j):
        axis_maps_from_attention_steps(self.steps[j % 4], (self.H, self.W), (self.Ho, self.Wo), maps_out=self.maps[j % 4])

    def _resample(self, j):
        cu.remap_separable(self.images[j % self.n], *self.maps[j % 4], mode=self.mode, channels_last=self.channels_last,
                           out=self.outs[j % self.n])

    def reset(self):
        self.k = 0

    def prime(self):
        k = self.k
        for j in range(k, k + 4):
            self._reduce(j)
        self._maps(k); self._maps(k + 1)

    def _graph(self, pairs):
        import math
        key = (self.k % math.lcm(4, self.n), pairs)
        if key not in self._graphs:
            g = torch.cuda.CUDAGraph()
            main = torch.cuda.Stream(device=self._dev)
            main.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(main):
                with torch.cuda.graph(g, stream=main):
                    for q in range(pairs):
                        self._launch(self.k + 2 * q)
            torch.cuda.current_stream().wait_stream(main)
            self._graphs[key] = g
        return self._graphs[key]

    def step(self):
        """One launch = two batches."""
        self._graph(1).replay()
        self.k += 2

    def run(self, n_batches: int, unroll_pairs: int = 4):
        """n_batches (even) batches: graphs of ``unroll_pairs`` launches, then single launches."""
        if n_batches % 2:
            raise ValueError("PairedStepWarp.run: an even number of batches")
        pairs = n_batches // 2
        while pairs >= unroll_pairs:
            self._graph(unroll_pairs).replay()
            self.k += 2 * unroll_pairs
            pairs -= unroll_pairs
        for _ in range(pairs):
            self.step()

    def tail(self):
        """The last four batches of a stream that ends: R(k), R(k+1), M + R of k+2, k+3 (their reduce is done)."""
        k = self.k
        self._resample(k); self._resample(k + 1)
        self._maps(k + 2); self._maps(k + 3)
        self._resample(k + 2); self._resample(k + 3)
        self.k += 4


def pair_slots(ring):
    """Views of a ring of equally shaped [B, ...] buffers as a ring of half the length of [2B, ...] buffers: slot i = the
    batches 2i and 2i+1 as ONE batch, for running two batches of a stream per launch (``MaskChainStream(pair_slots(images),
    pair_slots(masks), ...)``: a launch's ramp and tail, and at small batches its unfilled machine, are then paid once per
    two batches; the reference's BATCH_SIZE = 32 is a memory choice of the LLaVA pass, the warp is free to take two).
    Needs an even number of slots with slot 2i+1 directly behind slot 2i in memory, e.g. ``ring = list(torch.empty(n, B,
    ...))``.  The outputs of such a stream are [2B, ...] too: ``out[:B]`` / ``out[B:]``."""
    ring = list(ring)
    if len(ring) % 2:
        raise ValueError("pair_slots: an even number of slots expected")
    out = []
    for a, b in zip(ring[0::2], ring[1::2]):
        if (a.shape != b.shape or a.dtype != b.dtype or a.device != b.device or not (a.is_contiguous() and b.is_contiguous())
                or b.data_ptr() != a.data_ptr() + a.numel() * a.element_size()
                or a.untyped_storage().data_ptr() != b.untyped_storage().data_ptr()):
            raise ValueError("pair_slots: slot 2i+1 must lie directly behind slot 2i in the same allocation "
                             "(allocate the ring as one tensor [n, B, ...])")
        out.append(torch.as_strided(a, (2 * a.shape[0],) + tuple(a.shape[1:]), a.stride(), a.storage_offset()))
    return out


class MaskChainStream:
    """Steady-state form of :func:`warp_from_masks` -- the chain ``main_batched.py:243-287`` runs per image -- for a STREAM
    of equally shaped batches.  The stages of the chain belong to different batches and are independent:

        V(j)  revise_mask            masks[j] [B,24,24] float32       -> rev               llava.py:223-238
        L(j)  x255, PIL LANCZOS      rev                              -> mota [B,H,W] u8   llava.py:192-196,243,253
        P(j)  float64 marginals      mota                             -> axis sums         new_method.py:206-226
        F(j)  cumsum / np.interp     axis sums                        -> maps              new_method.py:228-265
        R(j)  uint8 cv2.remap        images[j] [B,H,W,3] + maps       -> outs[j]           new_method.py:268-271

    V, L, P, F are bound by instruction issue (integer multiply-adds, float64 adds in numpy's orders) and by dependent
    latency chains; R is the one that streams HBM.  Run back to back for every batch (``pattern="serial"``, the launches
    of ``warp_from_masks`` on static buffers) nothing overlaps.  ``pattern``:

        "branches": R(k) | P(k+1) -> F(k+1) | V(k+2) -> L(k+2) as three branches of one HIP graph, forked and joined every
                    step (existing kernels, the hardware schedules workgroups of the branches side by side)
        "fused":    R(k) | F(k+1) | P(k+2) | L(k+3) | V(k+4) as block ranges of ONE launch per step
                    (attwarp_mask_chain_step), replayed as graphs of ``unroll`` steps
        "ragged":   the same one launch per step through attwarp_mask_chain_ragged (every image described by a table entry):
                    equally sized images whose shape the uniform step refuses -- a width that is not a multiple of 4
        "auto":     "fused" when the shapes are eligible, else "ragged" when the ragged kernel takes them, else "branches"

    ``images`` / ``masks`` are rings of n static buffers (lists of equal length; a single tensor = a ring of one): batch k
    lives in slot k % n and is warped into ``outs[k % n]``.  Usage (d = pipeline depth: 2 for "branches", 4 for "fused"):

        mc = MaskChainStream(images, masks, out_size=(500, 500))
        masks[0 .. d-1] <- the first d batches;  mc.prime()
        for k in 0 .. N-1:
            images[k % n] <- batch k;  masks[(k + d) % n] <- mask of batch k + d (if any)
            out = mc.step()                                   # -> outs[k % n]

    Every step is bit-identical to ``warp_from_masks`` on its batch (same arithmetic per stage; the parity tests compare
    both with the oracle)."""

    PATTERNS = ("auto", "serial", "branches", "fused", "ragged")

    def __init__(self, images, masks, out_size=(500, 500), enhance_coe=10, kernel_size=3, transform="identity",
                 exp_scale=1.0, exp_divisor=1.0, apply_inverse=False, mode: str = "cv2", pattern: str = "auto"):
        if pattern not in self.PATTERNS:
            raise ValueError(f"MaskChainStream: pattern must be one of {self.PATTERNS}")
        self.images = list(images) if isinstance(images, (list, tuple)) else [images]
        self.masks = list(masks) if isinstance(masks, (list, tuple)) else [masks]
        if len(self.images) != len(self.masks) or not self.images:
            raise ValueError("MaskChainStream: images and masks must be rings of the same length")
        dev = require_gpu(*self.images, *self.masks)
        img0, m0 = self.images[0], self.masks[0]
        if any(i.dtype != torch.uint8 or i.dim() != 4 or i.shape != img0.shape or not i.is_contiguous() for i in self.images):
            raise TypeError("MaskChainStream: images must be contiguous uint8 [B,H,W,C] tensors of one shape")
        if any(m.dtype != torch.float32 or m.dim() != 3 or m.shape != m0.shape or not m.is_contiguous() for m in self.masks):
            raise TypeError("MaskChainStream: masks must be contiguous float32 [B,n,n] tensors of one shape")
        B, H, W, C = img0.shape
        if m0.shape[0] != B or m0.shape[1] != m0.shape[2]:
            raise ValueError("MaskChainStream: masks [B,n,n] must match the image batch")
        if kernel_size % 2 != 1:
            raise ValueError("MaskChainStream: kernel_size must be odd")
        self.n = len(self.images)
        self.B, self.H, self.W, self.C, self.g = B, H, W, C, int(m0.shape[1])
        self.Ho, self.Wo = int(out_size[0]), int(out_size[1])
        self.mode = mode
        self.enhance_coe, self.kernel_size = float(enhance_coe), int(kernel_size)
        self.transform = transform if transform in nm._KNOWN else "identity"
        self.exp_scale, self.exp_divisor, self.apply_inverse = float(exp_scale), float(exp_divisor), bool(apply_inverse)
        self._dev = dev
        lib = _lib.load()
        self.outs = [torch.empty(B, self.Ho, self.Wo, C, device=dev, dtype=torch.uint8) for _ in range(self.n)]
        self.k = 0
        self._graphs = {}
        self._side = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        self.pattern = pattern
        # the table of the transformed byte values the one-launch steps read for sqrt / exp / log (a constant of the stream)
        self._lut = _tables.attention_transform_lut(self.transform, self.exp_scale, self.exp_divisor, dev)
        if pattern == "ragged":               # (the uniform path's intermediates are not needed: every batch brings its own)
            if not (self.mode == "cv2" and ragged_eligible(H, W, C, self.g) and self.Wo * C <= RAGGED_MAX_ROW_BYTES):
                raise _lib.AttWarpError("MaskChainStream: this shape / these options do not run on the ragged chain kernel")
            self._build_ragged()
        else:
            # parity-indexed intermediates (batch j uses slot j % 2 of each)
            self.rev = [torch.empty(B, self.g, self.g, device=dev, dtype=torch.float32) for _ in (0, 1)]
            self.mota = [torch.empty(B, H, W, device=dev, dtype=torch.uint8) for _ in (0, 1)]
            self.ws = [torch.empty(lib.attwarp_axis_sums_workspace_bytes(B, H, W), device=dev, dtype=torch.uint8) for _ in (0, 1)]
            self.maps = [(torch.empty(B, self.Wo, device=dev, dtype=torch.float32),
                          torch.empty(B, self.Ho, device=dev, dtype=torch.float32)) for _ in (0, 1)]
            self._tmp = torch.empty(B, self.g, W, device=dev, dtype=torch.uint8)       # two-pass Lanczos forms only
            self._tx = _tables.lanczos_tables(self.g, W, dev) if W != self.g else (None, None, 0)
            self._ty = _tables.lanczos_tables(self.g, H, dev) if H != self.g else (None, None, 0)
            if pattern in ("auto", "fused"):
                try:
                    self._fused_launch(0, dry=True)
                    self.pattern = "fused"
                except (_lib.AttWarpError, AttributeError):
                    if pattern == "fused":
                        raise
                    self.pattern = "branches"
            if pattern == "auto" and self.pattern == "branches":
                if self.mode == "cv2" and ragged_eligible(H, W, C, self.g) and self.Wo * C <= RAGGED_MAX_ROW_BYTES:
                    self.pattern = "ragged"
                    del self.rev, self.mota, self.ws, self.maps, self._tmp
                    self._build_ragged()
        self.depth = {"serial": 0, "branches": 2, "fused": 4, "ragged": 4}[self.pattern]

    def _build_ragged(self):
        """Every (ring slot, batch parity) RaggedBatch of the stream, built HERE -- before any capture: building one plans
        a table on the host and uploads it, which must never become a node of a HIP graph (it would be re-executed, from a
        staging buffer that has been reused since, on every replay).  Batch j of the stream lives in ring slot j % n and
        uses the intermediates of parity j & 1: one RaggedBatch per (slot, parity) that occurs -- n for an even ring, 2 n for
        an odd one -- each with its own table (the table names the slot's images), all batches of one parity SHARING one set
        of intermediates (rev / mota / sums / maps: two sets in all, as the uniform step)."""
        import math
        self._rb = {}
        first = {}
        for j in range(math.lcm(2, self.n)):
            key = (j % self.n, j & 1)
            img = self.images[key[0]]
            rb = RaggedBatch([img[b] for b in range(self.B)], (self.Ho, self.Wo), self.g, out=self.outs[key[0]],
                             share=first.get(key[1]))
            rb.masks = self.masks[key[0]]
            first.setdefault(key[1], rb)
            self._rb[key] = rb

    # ---- the stages on static buffers (same entry points as warp_from_masks) ----
    def _V(self, j):
        with torch.cuda.device(self._dev):
            call("attwarp_mask_postproc", ptr(self.masks[j % self.n]), self.B, self.g, self.kernel_size, self.enhance_coe,
                 ptr(self.rev[j & 1]), stream_ptr(self._dev))

    def _L(self, j):
        (bx, kx, ksx), (by, ky, ksy) = self._tx, self._ty
        with torch.cuda.device(self._dev):
            call("attwarp_mask_upsample_lanczos", ptr(self.rev[j & 1]), None, self.B, self.g, self.g, self.H, self.W,
                 ptr(bx), ptr(kx), ksx, ptr(by), ptr(ky), ksy, ptr(self._tmp), ptr(self.mota[j & 1]), stream_ptr(self._dev))

    def _PF(self, j):
        mx, my = self.maps[j & 1]
        with torch.cuda.device(self._dev):
            call("attwarp_axis_maps_from_attention", ptr(self.mota[j & 1]), _lib.U8, self.B, self.H, self.W, self.Wo, self.Ho,
                 _lib.TRANSFORM_IDS[self.transform], self.exp_scale, self.exp_divisor, int(self.apply_inverse), ptr(mx),
                 ptr(my), ptr(self.ws[j & 1]), stream_ptr(self._dev))

    def _R(self, j):
        cu.remap_separable(self.images[j % self.n], *self.maps[j & 1], mode=self.mode, channels_last=True,
                           out=self.outs[j % self.n])

    def _fused_launch(self, k, dry=False):
        """R(k) | F(k+1) | P(k+2) | L(k+3) | V(k+4) as ONE launch (attwarp_mask_chain_step).  ``dry``: the eligibility
        trial of the constructor -- the stages write the buffers of the parities they would write in step k, from whatever
        the inputs hold; prime() overwrites all of it."""
        if self.mode != "cv2":
            raise _lib.AttWarpError("mask chain step: cv2 mode only (the integer uint8 resample; what both reference drivers compute)")
        (bx, kx, ksx), (by, ky, ksy) = self._tx, self._ty
        if bx is None or by is None:
            raise _lib.AttWarpError("mask chain step: the mask must be up-sampled on both axes")
        p, q = k & 1, (k + 1) & 1
        mx, my = self.maps[p]
        nx, ny = self.maps[q]
        with torch.cuda.device(self._dev):
            call("attwarp_mask_chain_step", ptr(self.images[k % self.n]), ptr(self.outs[k % self.n]), self.B, self.C, self.H,
                 self.W, self.Ho, self.Wo, ptr(mx), ptr(my),
                 ptr(self.ws[q]), ptr(nx), ptr(ny),                      # F(k+1)
                 ptr(self.mota[p]), ptr(self.ws[p]),                      # P(k+2)
                 ptr(self.rev[q]), ptr(bx), ptr(kx), ksx, ptr(by), ptr(ky), ksy, ptr(self.mota[q]),   # L(k+3)
                 ptr(self.masks[(k + 4) % self.n]), self.g, self.kernel_size, self.enhance_coe, ptr(self.rev[p]),   # V(k+4)
                 _lib.TRANSFORM_IDS[self.transform], self.exp_scale, self.exp_divisor, int(self.apply_inverse), ptr(self._lut),
                 stream_ptr(self._dev))

    def _ragged(self, j):
        """Batch j of the stream: the RaggedBatch of slot j % n (its images, masks and output buffer) with the
        intermediates of parity j & 1 (all built by the constructor)."""
        return self._rb[(j % self.n, j & 1)]

    def _ragged_launch(self, **stages):
        ragged_chain_launch(**{s: self._ragged(j) for s, j in stages.items()}, enhance_coe=self.enhance_coe, kernel_size=self.kernel_size,
                            transform=self.transform, exp_scale=self.exp_scale, exp_divisor=self.exp_divisor,
                            apply_inverse=self.apply_inverse)

    # ---- driving ----
    def prime(self):
        """Everything the first step expects to find done for the batches at ring positions k .. k+depth-1 (serial)."""
        k = self.k
        if self.pattern == "ragged":
            for j, stages in ((k, "VLPF"), (k + 1, "VLP"), (k + 2, "VL"), (k + 3, "V")):
                for st in stages:
                    self._ragged_launch(**{st: j})
            return
        if self.pattern == "branches":      # step k runs R(k) | PF(k+1) | VL(k+2)
            self._V(k); self._L(k); self._PF(k)
            self._V(k + 1); self._L(k + 1)
        elif self.pattern == "fused":       # step k runs R(k) | F(k+1) | P(k+2) | L(k+3) | V(k+4)
            for j in (k, k + 1, k + 2):
                self._V(j); self._L(j)
                if j <= k + 1:
                    self._PF(j)             # (P(k+1) included: its F runs again inside step k, same result)
            self._V(k + 3)

    def _one(self, k):
        if self.pattern == "serial":
            self._V(k); self._L(k); self._PF(k); self._R(k)
        elif self.pattern == "fused":
            self._fused_launch(k)
        elif self.pattern == "ragged":
            self._ragged_launch(R=k, F=k + 1, P=k + 2, L=k + 3, V=k + 4)
        else:
            raise RuntimeError("branches are only defined inside a capture")

    def _capture(self, k, unroll):
        g = torch.cuda.CUDAGraph()
        main = torch.cuda.Stream(device=self._dev)
        main.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(main):
            with torch.cuda.graph(g, stream=main):
                for u in range(unroll):
                    if self.pattern == "branches":
                        s1, s2 = self._side
                        s1.wait_stream(main); s2.wait_stream(main)          # fork
                        with torch.cuda.stream(s1):
                            self._PF(k + 1)
                        with torch.cuda.stream(s2):
                            self._V(k + 2); self._L(k + 2)
                        self._R(k)
                        main.wait_stream(s1); main.wait_stream(s2)          # join
                    else:
                        self._one(k)
                    k += 1
        torch.cuda.current_stream().wait_stream(main)
        return g

    def _graph(self, unroll):
        # the buffer indices of a step depend on k mod n (ring) and k mod 2 (parity)
        import math
        key = (self.k % math.lcm(2, self.n), unroll)
        if key not in self._graphs:
            self._graphs[key] = self._capture(self.k, unroll)
        return self._graphs[key]

    @property
    def out(self) -> torch.Tensor:
        """Output buffer of the most recent step."""
        return self.outs[(self.k - 1) % self.n]

    def step(self) -> torch.Tensor:
        self._graph(1).replay()
        self.k += 1
        return self.out

    def run(self, n: int, unroll: int = 8) -> torch.Tensor:
        """n steps on the buffers as they are (steady-state measurement / a producer that stays ahead of the ring): graphs
        of ``unroll`` steps (one host call each), then single steps.  A graph is keyed by the ring position it starts at
        (mod lcm(2, ring length)), so a stream that is reset() and run again replays the graphs of the first pass."""
        unroll = max(2, unroll - unroll % 2)
        while n >= unroll:
            self._graph(unroll).replay()
            self.k += unroll
            n -= unroll
        for _ in range(n):
            self.step()
        return self.out

    def reset(self):
        self.k = 0

    def drain(self) -> torch.Tensor:
        """The last ``depth`` batches of a stream that ends (no further masks): the remaining stages, serially."""
        k = self.k
        if self.pattern == "ragged":
            for j, stages in ((k, "R"), (k + 1, "FR"), (k + 2, "PFR"), (k + 3, "LPFR")):
                for st in stages:
                    self._ragged_launch(**{st: j})
        elif self.pattern == "branches":
            self._R(k); self._PF(k + 1); self._R(k + 1)
        elif self.pattern == "fused":
            self._R(k)
            self._PF(k + 1); self._R(k + 1)
            self._PF(k + 2); self._R(k + 2)
            self._L(k + 3); self._PF(k + 3); self._R(k + 3)
        self.k += self.depth
        return self.out


# ---- the main_batched chain for batches of DIFFERENTLY sized images -------------------------------------------------------
RAGGED_MAX_ROW_BYTES = 4096          # attwarp_ragged_plan: rows of the integer cv2 resample
RAGGED_MAX_SIDE = 8192


def ragged_eligible(h: int, w: int, c: int, g: int = GRID) -> bool:
    """Does an [h,w,c] uint8 image run on the ragged chain kernel (attwarp.h: limits of attwarp_ragged_plan)?"""
    return 4 <= w * c <= RAGGED_MAX_ROW_BYTES and h > g and w > g and max(h, w) <= RAGGED_MAX_SIDE and c <= 4


_RAGGED_IMAGE_DTYPE = np.dtype([("image", "<u8"), ("bounds_x", "<u8"), ("kk_x", "<u8"), ("bounds_y", "<u8"), ("kk_y", "<u8"),
                                ("H", "<i4"), ("W", "<i4"), ("ksize_x", "<i4"), ("reserved", "<i4")])      # attwarp_ragged_image
_AXIS_TABLES: dict = {}
_STAGING: dict = {}                   # device index -> pinned staging buffers: [tensor, event of the last copy out of it, free]
_STAGING_LOCK = __import__("threading").Lock()     # (host threads on their own streams share the pool)


_AXIS_TABLES_MAX = 4096              # distinct (device, g, size) entries kept (40 KB of device memory each at size 1024)


def _axis_tables(g: int, n: int, dev: torch.device):
    """(bounds pointer, coefficient pointer, ksize, bounds tensor, coefficient tensor) of Pillow's g -> n LANCZOS tables on
    `dev`, cached per size (bounded: the oldest entries go first; a RaggedBatch keeps the tensors it points to alive)."""
    key = (dev.index, g, n)
    t = _AXIS_TABLES.get(key)
    if t is None:
        b, k, ks = _tables.lanczos_tables(g, n, dev)
        t = (b.data_ptr(), k.data_ptr(), ks, b, k)
        with _STAGING_LOCK:                   # (host threads share the cache: insert + evict under one lock)
            while len(_AXIS_TABLES) >= _AXIS_TABLES_MAX:
                _AXIS_TABLES.pop(next(iter(_AXIS_TABLES)), None)
            _AXIS_TABLES[key] = t
    return t


def _upload(host: "np.ndarray", dst: torch.Tensor):
    """host bytes -> dst (device uint8), asynchronously on the current stream of dst's device through a pooled pinned buffer
    (one pool per device: a slot's event is recorded on, and queried against, the stream that carries its copy).
    Refuses to run inside a stream capture: the copy would become a graph node and be re-executed on every replay from a
    staging buffer that has long been reused."""
    dev = dst.device
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("attwarp_amd: a host-to-device table upload inside a stream capture (build RaggedBatch objects / call "
                           "upload_images before capturing; a captured copy would replay from a reused staging buffer)")
    n = host.size
    with torch.cuda.device(dev):
        with _STAGING_LOCK:
            pool = _STAGING.setdefault(dev.index, [])
            slot = None
            for s in pool:
                if s[0].numel() >= n and s[2] and s[1].query():
                    slot = s
                    break
            if slot is None:
                if len(pool) >= 64:               # (only if nothing ever completes: keep the pool bounded)
                    torch.cuda.synchronize(dev)
                    del pool[1:]
                slot = [torch.empty(max(n, 1 << 16), dtype=torch.uint8).pin_memory(), torch.cuda.Event(), True]
                pool.append(slot)
            slot[2] = False                       # taken: nobody else may pick it until its copy has been enqueued and recorded
        try:
            slot[0][:n].numpy()[:] = host
            dst.copy_(slot[0][:n], non_blocking=True)
            slot[1].record(torch.cuda.current_stream(dev))
        finally:
            slot[2] = True


class RaggedBatch:
    """One batch of differently sized uint8 images [H_i,W_i,C] on the GPU, planned for ``attwarp_mask_chain_ragged``:
    the table (host copy in pinned memory + device copy), and the batch's own intermediates and outputs --
    ``rev`` [B,g,g], ``mota`` (the up-sampled masks, packed), ``sums`` (axis sums), ``maps`` and ``out`` [B,H_out,W_out,C].
    The images are referenced, not copied: keep them alive and unchanged until the batch's resample has run."""

    def __init__(self, images, out_size=(500, 500), g: int = GRID, out: Optional[torch.Tensor] = None,
                 share: Optional["RaggedBatch"] = None):
        """``share``: a RaggedBatch of the SAME image sizes whose intermediates (rev / mota / sums / maps) this batch adopts
        instead of allocating its own -- the batches of one parity of a stream never have two stages in flight at once."""
        import ctypes
        images = list(images)
        if not images:
            raise ValueError("RaggedBatch: empty batch")
        dev = require_gpu(*images)
        C = int(images[0].shape[2]) if images[0].dim() == 3 else -1
        for im in images:
            if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != C or not im.is_contiguous():
                raise TypeError("RaggedBatch: images must be contiguous uint8 [H,W,C] tensors with one channel count")
        self.images = images
        self.B, self.C, self.g = len(images), C, int(g)
        self.Ho, self.Wo = int(out_size[0]), int(out_size[1])
        self._dev = dev
        lib = _lib.load()
        # the caller-side records (attwarp_ragged_image) as one structured array: pointers and sizes column by column
        rec = np.zeros(self.B, dtype=_RAGGED_IMAGE_DTYPE)
        rec["image"] = [im.data_ptr() for im in images]
        rec["H"] = hs = [int(im.shape[0]) for im in images]
        rec["W"] = ws = [int(im.shape[1]) for im in images]
        tx = [_axis_tables(self.g, w, dev) for w in ws]
        ty = [_axis_tables(self.g, h, dev) for h in hs]
        if any(t[2] != 8 for t in ty):
            raise _lib.AttWarpError("RaggedBatch: an image is too low for the 8-tap vertical mask up-sampling (H < 24)")
        rec["bounds_x"] = [t[0] for t in tx]; rec["kk_x"] = [t[1] for t in tx]; rec["ksize_x"] = [t[2] for t in tx]
        rec["bounds_y"] = [t[0] for t in ty]; rec["kk_y"] = [t[1] for t in ty]
        self._keep = [t[3:] for t in tx] + [t[3:] for t in ty]      # the coefficient tensors the table points to
        rp = ctypes.c_void_p(rec.ctypes.data)
        nbytes = lib.attwarp_ragged_table_bytes(rp, self.B, C, self.g, self.Ho, self.Wo)
        if nbytes == 0:
            raise _lib.AttWarpError("attwarp_ragged_table_bytes: " + lib.attwarp_last_error().decode("utf-8", "replace"))
        # host copy: a plain buffer this batch keeps (the entry point reads the header from it at every launch); the upload
        # goes through a pooled pinned staging buffer (allocating pinned memory per batch costs more than the five launches)
        self.table_host = np.empty(nbytes, dtype=np.uint8)
        call("attwarp_ragged_plan", rp, self.B, C, self.g, self.Ho, self.Wo, self.table_host.ctypes.data, nbytes)
        self.header = _lib.RaggedHeader.from_address(self.table_host.ctypes.data)
        self.table_dev = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        _upload(self.table_host, self.table_dev)
        h = self.header
        if share is not None:
            if (share.B, share.g, share.Ho, share.Wo, int(share.header.mota_bytes), int(share.header.sums_bytes)) != \
                    (self.B, self.g, self.Ho, self.Wo, int(h.mota_bytes), int(h.sums_bytes)) or share._dev != dev:
                raise ValueError("RaggedBatch: `share` must be a batch of the same image sizes on the same device")
            self.rev, self.mota, self.sums, self.map_x, self.map_y = share.rev, share.mota, share.sums, share.map_x, share.map_y
        else:
            self.rev = torch.empty(self.B, self.g, self.g, device=dev, dtype=torch.float32)
            self.mota = torch.empty(int(h.mota_bytes), device=dev, dtype=torch.uint8)
            self.sums = torch.empty(int(h.sums_bytes) // 8, device=dev, dtype=torch.float64)
            self.map_x = torch.empty(self.B, self.Wo, device=dev, dtype=torch.float32)
            self.map_y = torch.empty(self.B, self.Ho, device=dev, dtype=torch.float32)
        if out is None:
            out = torch.empty(self.B, self.Ho, self.Wo, C, device=dev, dtype=torch.uint8)
        elif tuple(out.shape) != (self.B, self.Ho, self.Wo, C) or out.dtype != torch.uint8 or not out.is_contiguous() or out.device != dev:
            raise ValueError("RaggedBatch: out must be a dense uint8 [B,H_out,W_out,C] tensor on the images' device")
        self.out = out
        self.masks = None                   # [B,g,g] float32: set by the caller before V runs

    @property
    def algorithmic_bytes(self) -> int:
        """Chain-level bytes of this batch: mask written + read (2 H W), image read (C H W), output written."""
        px = sum(int(i.shape[0]) * int(i.shape[1]) for i in self.images)
        return px * (2 + self.C) + self.B * self.Ho * self.Wo * self.C

    def mota_of(self, b: int) -> torch.Tensor:
        """The up-sampled uint8 mask [H_b,W_b] of image b (what blend_mask returns as ``mota_mask``, llava.py:253)."""
        import ctypes
        rec = (_RaggedImageDev * self.B).from_address(self.table_host.ctypes.data + int(self.header.off_images))
        H, W = int(self.images[b].shape[0]), int(self.images[b].shape[1])
        off = int(rec[b].mota_off)
        return self.mota[off:off + H * W].view(H, W)


class _RaggedImageDev(__import__("ctypes").Structure):
    """The per-image record inside a table (chain_ragged.hip: RaggedImage); read here only for ``mota_off``."""
    import ctypes as _c
    _fields_ = [("image", _c.c_void_p), ("bounds_x", _c.c_void_p), ("kk_x", _c.c_void_p), ("bounds_y", _c.c_void_p),
                ("kk_y", _c.c_void_p), ("H", _c.c_int32), ("W", _c.c_int32), ("ksize_x", _c.c_int32), ("plan_w", _c.c_int32),
                ("plan_h", _c.c_int32), ("l_nchunks", _c.c_int32), ("l_rows_per_chunk", _c.c_int32), ("ki", _c.c_int32),
                ("mota_off", _c.c_int64), ("sums_off", _c.c_int64)]


def upload_images(images, device=None):
    """A batch of host images as the reference's driver holds them (`b_images`: PIL images or uint8 [H,W,C] arrays of
    different sizes, main_batched.py:246) -> list of uint8 [H_i,W_i,C] GPU tensors: packed back to back into ONE staging
    buffer and copied with ONE host-to-device transfer (the ragged kernels take images at any byte address, so the views
    need no padding).  PIL images are taken as RGB; the warp does not depend on the channel order."""
    dev = torch.device(device) if device is not None else _lib.default_device()
    arrs = []
    for im in images:
        a = np.asarray(im.convert("RGB")) if hasattr(im, "convert") else np.asarray(im)
        if a.dtype != np.uint8 or a.ndim not in (2, 3):
            raise TypeError("upload_images: uint8 [H,W] / [H,W,C] arrays or PIL images expected")
        arrs.append(a[:, :, None] if a.ndim == 2 else a)
    sizes = [a.size for a in arrs]
    total = int(sum(sizes))
    host = np.empty(total, dtype=np.uint8)
    off = 0
    for a, n in zip(arrs, sizes):
        host[off:off + n] = a.reshape(-1)
        off += n
    flat = torch.empty(total, device=dev, dtype=torch.uint8)
    with torch.cuda.device(dev):
        _upload(host, flat)
    out, off = [], 0
    for a, n in zip(arrs, sizes):
        out.append(flat[off:off + n].view(*a.shape))
        off += n
    return out


def ragged_chain_launch(R: Optional[RaggedBatch] = None, F: Optional[RaggedBatch] = None, P: Optional[RaggedBatch] = None,
                        L: Optional[RaggedBatch] = None, V: Optional[RaggedBatch] = None, enhance_coe=10, kernel_size=3,
                        transform="identity", exp_scale=1.0, exp_divisor=1.0, apply_inverse=False):
    """ONE launch of ``attwarp_mask_chain_ragged``: the resample of batch R, the map construction of F, the marginals of P,
    the mask up-sampling of L and revise_mask of V (``V.masks`` [B,g,g] float32) -- five different batches of a stream, or
    any subset (one batch alone: five launches, V -> L -> P -> F -> R).  ``transform`` .. ``apply_inverse``:
    save_warped_image's keyword arguments (new_method.py:405-411), used by P and F; unknown names mean identity (:400-403)."""
    some = next(b for b in (R, F, P, L, V) if b is not None)
    dev = some._dev
    if transform not in nm._KNOWN:
        transform = "identity"
    lut = _tables.attention_transform_lut(transform, exp_scale, exp_divisor, dev) if P is not None else None
    def tab(b):
        return (None, None) if b is None else (b.table_host.ctypes.data, ptr(b.table_dev))
    if V is not None and (V.masks is None or V.masks.dtype != torch.float32 or tuple(V.masks.shape) != (V.B, V.g, V.g)
                          or not V.masks.is_contiguous() or V.masks.device != dev):
        raise ValueError("ragged_chain_launch: V.masks must be a dense float32 [B,g,g] tensor on the batch's device")
    with torch.cuda.device(dev):
        call("attwarp_mask_chain_ragged",
             *tab(R), ptr(R.out) if R else None, ptr(R.map_x) if R else None, ptr(R.map_y) if R else None,
             *tab(F), ptr(F.sums) if F else None, ptr(F.map_x) if F else None, ptr(F.map_y) if F else None,
             *tab(P), ptr(P.mota) if P else None, ptr(P.sums) if P else None,
             *tab(L), ptr(L.rev) if L else None, ptr(L.mota) if L else None,
             ptr(V.masks) if V else None, V.B if V else 0, some.g, int(kernel_size), float(enhance_coe), ptr(V.rev) if V else None,
             _lib.TRANSFORM_IDS[transform], float(exp_scale), float(exp_divisor), int(bool(apply_inverse)), ptr(lut),
             stream_ptr(dev))


def warp_from_masks_ragged(images, attn24: torch.Tensor, out_size=(500, 500), enhance_coe=10, kernel_size=3,
                           transform="identity", exp_scale=1.0, exp_divisor=1.0, apply_inverse=False, mode: str = "cv2",
                           return_batch: bool = False):
    """The ``main_batched.py:243-287`` chain for a batch of DIFFERENTLY sized images, as that driver holds them:
    ``images``: list of uint8 [H_i,W_i,C] GPU tensors (any sizes, any width -- 683 x 1024 like 1024 x 768); ``attn24``
    [B,24,24] -> dense uint8 [B,H_out,W_out,C].  Five launches for the whole batch (revise_mask, LANCZOS up-sampling to
    every image's own size, float64 marginals, CDF / np.interp maps, cv2 resample), each over all images at once.
    ``transform`` / ``exp_scale`` / ``exp_divisor`` / ``apply_inverse``: save_warped_image's own keyword arguments
    (new_method.py:405-411; the driver passes "identity", 1.0, 1.0, False); same signature as :func:`warp_from_masks`.
    Images outside the ragged kernel's limits (rows wider than 4096 bytes, sides <= 24 or > 8192) and ``mode="exact"``
    (the ragged resample is the integer cv2 one) run through :func:`warp_from_masks` one by one."""
    images = list(images)
    B = len(images)
    if B == 0:
        raise ValueError("warp_from_masks_ragged: empty batch")
    if attn24.dim() != 3 or attn24.shape[0] != B or attn24.shape[1] != attn24.shape[2]:
        raise ValueError("warp_from_masks_ragged: one square [g,g] map per image expected ([B,g,g])")
    dev = require_gpu(attn24, *images)
    if any(i.dim() != 3 or i.dtype != torch.uint8 for i in images):
        raise TypeError("warp_from_masks_ragged: images must be uint8 [H,W,C] tensors")
    C = int(images[0].shape[2])
    if any(int(i.shape[2]) != C for i in images):
        raise ValueError("warp_from_masks_ragged: the images of one batch must share their channel count "
                         f"(got {sorted({int(i.shape[2]) for i in images})}): the output is one dense [B,H_out,W_out,C] tensor")
    if mode not in ("cv2", "exact"):
        raise ValueError(f"warp_from_masks_ragged: unknown mode {mode!r}")
    g = int(attn24.shape[-1])
    Ho, Wo = int(out_size[0]), int(out_size[1])
    xf = dict(transform=transform, exp_scale=exp_scale, exp_divisor=exp_divisor, apply_inverse=apply_inverse)
    ok = [mode == "cv2" and ragged_eligible(int(i.shape[0]), int(i.shape[1]), C, g) for i in images]
    if Wo * C > RAGGED_MAX_ROW_BYTES:
        ok = [False] * B
    masks = attn24.float().contiguous()
    rb = None
    if all(ok):
        try:
            rb = RaggedBatch(images, out_size, g)
        except _lib.AttWarpError:          # a limit ragged_eligible does not model (e.g. the LDS of a very long axis): per image
            ok = [False] * B
    if rb is not None:
        rb.masks = masks
        for stage in "VLPFR":
            ragged_chain_launch(**{stage: rb}, enhance_coe=enhance_coe, kernel_size=kernel_size, **xf)
        return (rb.out, rb) if return_batch else rb.out
    out = torch.empty(B, Ho, Wo, C, device=dev, dtype=torch.uint8)
    idx = [b for b in range(B) if ok[b]]
    if idx:
        sub = warp_from_masks_ragged([images[b] for b in idx], masks[idx], out_size, enhance_coe, kernel_size, **xf)
        out[idx] = sub
    for b in range(B):
        if not ok[b]:
            out[b] = warp_from_masks(images[b][None], masks[b:b + 1], out_size, enhance_coe, kernel_size, mode=mode, **xf)[0]
    return (out, None) if return_batch else out


class RaggedMaskChainStream:
    """A stream of ragged batches, one launch per batch: R(k) | F(k+1) | P(k+2) | L(k+3) | V(k+4) of
    ``attwarp_mask_chain_ragged`` (the ragged twin of :class:`MaskChainStream`).

        st = RaggedMaskChainStream(out_size=(500, 500))
        for images, attn24 in batches:                 # lists of uint8 [H_i,W_i,3] GPU tensors, [B,24,24]
            done = st.push(images, attn24)             # -> the RaggedBatch whose resample was just enqueued (4 pushes back)
            if done is not None: use(done.out)
        for done in st.flush(): use(done.out)

    Every output equals :func:`warp_from_masks_ragged` on its batch, bit for bit (same stage bodies).  For steady-state
    measurements ``ring`` + ``run`` replay one launch per batch over prebuilt batches (optionally as HIP graphs)."""

    DEPTH = 4

    def __init__(self, out_size=(500, 500), enhance_coe=10, kernel_size=3, g: int = GRID, transform="identity", exp_scale=1.0,
                 exp_divisor=1.0, apply_inverse=False):
        self.out_size = (int(out_size[0]), int(out_size[1]))
        self.enhance_coe, self.kernel_size, self.g = float(enhance_coe), int(kernel_size), int(g)
        # save_warped_image's keyword arguments (new_method.py:405-411), constants of the stream
        self._xf = dict(transform=transform if transform in nm._KNOWN else "identity", exp_scale=float(exp_scale),
                        exp_divisor=float(exp_divisor), apply_inverse=bool(apply_inverse))
        self._q = []                       # the batches in flight, oldest first: [k-4 .. k]
        self._graphs = {}

    def _stages(self, **stages):
        ragged_chain_launch(**stages, enhance_coe=self.enhance_coe, kernel_size=self.kernel_size, **self._xf)

    def _launch(self, q):
        """q: the five batches [R, F, P, L, V] of this step (None where the stream has none)."""
        if any(b is not None for b in q):
            self._stages(R=q[0], F=q[1], P=q[2], L=q[3], V=q[4])

    def push(self, images, attn24: torch.Tensor, out: Optional[torch.Tensor] = None):
        rb = RaggedBatch(images, self.out_size, self.g, out=out)
        rb.masks = attn24.float().contiguous()
        self._q.append(rb)
        q = ([None] * (self.DEPTH + 1) + self._q)[-(self.DEPTH + 1):]
        self._launch(q)
        if len(self._q) > self.DEPTH:
            return self._q.pop(0)
        return None

    def flush(self):
        """The launches that finish the batches still in flight; returns them oldest first."""
        done = []
        # the youngest batch in flight has had V only, the one before it V and L, ...: one launch moves every batch one
        # stage on ([R, F, P, L, V] slots, oldest first)
        q = ([None] * (self.DEPTH + 1) + self._q)[-(self.DEPTH + 1):]
        q = q[1:] + [None]
        while any(b is not None for b in q):
            self._launch(q)
            if q[0] is not None:
                done.append(q[0])
            q = q[1:] + [None]
        self._q = []
        return done

    # ---- steady state over prebuilt batches (measurement, or a producer that stays ahead of the ring) ----
    def ring(self, batches):
        """Adopt n >= 5 prebuilt RaggedBatch objects (tables on the device, ``masks`` set) as a ring: step k runs
        R(k) | F(k+1) | P(k+2) | L(k+3) | V(k+4) on slots k % n .. (k+4) % n."""
        self._ring = list(batches)
        if len(self._ring) < self.DEPTH + 1:
            raise ValueError("RaggedMaskChainStream.ring: at least 5 batches")
        self.k = 0

    def prime(self):
        """Run V..F of the first four ring slots serially, so that step 0 finds what it expects."""
        r = self._ring
        for j, stages in ((0, "VLPF"), (1, "VLP"), (2, "VL"), (3, "V")):
            for s in stages:
                self._stages(**{s: r[j]})

    def _step(self, k):
        r, n = self._ring, len(self._ring)
        self._launch([r[(k + i) % n] for i in range(self.DEPTH + 1)])

    def run(self, n_steps: int, unroll: int = 0):
        """n_steps launches over the ring; unroll > 0: replayed as HIP graphs of ``unroll`` steps (one host call each)."""
        n = len(self._ring)
        if unroll > 0:
            unroll = n * max(1, unroll // n)            # a graph covers whole turns of the ring
            while n_steps >= unroll:
                key = (self.k % n, unroll)
                if key not in self._graphs:
                    self._graphs[key] = capture_steps(self._step, self.k, unroll, self._ring[0]._dev)
                self._graphs[key].replay()
                self.k += unroll
                n_steps -= unroll
        for _ in range(n_steps):
            self._step(self.k)
            self.k += 1

    def drain_ring(self):
        """Finish the four batches in flight behind the last step, serially (their outputs then equal the stream's)."""
        r, n, k = self._ring, len(self._ring), self.k
        for j, stages in ((k, "R"), (k + 1, "FR"), (k + 2, "PFR"), (k + 3, "LPFR")):
            for s in stages:
                self._stages(**{s: r[j % n]})
        self.k += self.DEPTH


def capture_steps(step_fn, k0: int, n: int, dev: torch.device):
    """Capture step_fn(k0) .. step_fn(k0 + n - 1) (launches on the current stream, static buffers) as one HIP graph."""
    g = torch.cuda.CUDAGraph()
    main = torch.cuda.Stream(device=dev)
    main.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(main):
        with torch.cuda.graph(g, stream=main):
            for u in range(n):
                step_fn(k0 + u)
    torch.cuda.current_stream().wait_stream(main)
    return g
