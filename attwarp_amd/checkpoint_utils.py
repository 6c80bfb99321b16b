"""Drop-in for the hot-path half of ``model/marginalnet_full_dataset/checkpoint_utils.py``
(reference lines 17-204).  Same names, argument meaning and error behaviour; tensors in ->
tensors out on the SAME device and dtype as the input (reference contract, :203).

Differences by design (MI355X-first):
* nothing leaves the GPU: the reference copies the whole batch to the host, loops over samples
  in Python and calls ``cv2.remap`` (:152-204); here one kernel builds all inverse maps and one
  resamples the whole batch;
* ``upsample_pdf_right_inverse`` applies a cached 24x24 inverse instead of rebuilding the pooling
  matrix with 48 host syncs and an LU per call (:104-121);
* the resample arithmetic defaults to what ``cv2.remap(INTER_LINEAR)`` computes (``mode="cv2"``: coordinates
  rounded to 1/32 pixel, 4 table weights for float32, 15-bit fixed point for uint8 -- OpenCV's published algorithm);
  ``mode="exact"`` is bilinear on the unquantised coordinates (= ``F.grid_sample(bilinear, border,
  align_corners=True)``).  Both run on the same staged kernels.  See DESIGN.md "parity".
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib, _tables
from ._lib import call, ptr, require_gpu, stream_ptr


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


def _make_strictly_increasing(Fcdf: torch.Tensor, eps: float = 1e-4) -> torch.Tensor:
    """Reference :17-28.  Fcdf (B,N) -> repaired CDF (B,N) float32.  (Forward-only kernel; when autograd has to see
    through it the same steps run on differentiable stock ops, see ``_lib.needs_grad``.)"""
    dev = require_gpu(Fcdf)
    if _lib.needs_grad(Fcdf):
        Fn = torch.cummax(torch.nan_to_num(Fcdf, nan=0.0, posinf=1.0, neginf=0.0), dim=1)[0]
        d = (Fn[:, 1:] - Fn[:, :-1]).clamp(min=eps / max(Fn.shape[1], 1))
        Fx = torch.cat([Fn[:, :1], Fn[:, :1] + torch.cumsum(d, dim=1)], dim=1)
        Fx = (Fx / Fx[:, -1:].clamp_min(1e-6)).clamp(0.0, 1.0)
        return torch.cat([Fx[:, :-1], torch.ones_like(Fx[:, -1:])], dim=1)
    F = _f32c(Fcdf)
    B, N = F.shape
    out = torch.empty_like(F)
    with torch.cuda.device(dev):
        call("attwarp_make_strictly_increasing", ptr(F), B, N, float(eps), ptr(out), stream_ptr(dev))
    return out


def cdf_from_density(p: torch.Tensor) -> torch.Tensor:
    """Reference :30-41.  p (B,N) -> non-decreasing CDF in [0,1] ending at 1 (float32)."""
    dev = require_gpu(p)
    if _lib.needs_grad(p):
        q = torch.nan_to_num(p.float().clamp_min(0), nan=0.0, posinf=0.0, neginf=0.0)
        Fp = torch.cumsum(q / q.sum(dim=1, keepdim=True).clamp_min(1e-6), dim=1)
        return torch.cat([Fp[:, :-1], torch.ones_like(Fp[:, -1:])], dim=1)
    x = _f32c(p)
    B, N = x.shape
    out = torch.empty_like(x)
    with torch.cuda.device(dev):
        call("attwarp_cdf_from_density", ptr(x), B, N, ptr(out), stream_ptr(dev))
    return out


def gt_marginals(A: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Reference :43-51.  A (B,1,H,W) -> (px (B,W), py (B,H)), normalised."""
    dev = require_gpu(A)
    if _lib.needs_grad(A):
        Ap = A.clamp_min(0)
        mx, my = Ap.sum(dim=2).squeeze(1), Ap.sum(dim=3).squeeze(1)
        return mx / mx.sum(dim=1, keepdim=True).clamp_min(1e-6), my / my.sum(dim=1, keepdim=True).clamp_min(1e-6)
    B, _, H, W = A.shape
    x = _f32c(A[:, 0])
    px = torch.empty(B, W, device=dev, dtype=torch.float32)
    py = torch.empty(B, H, device=dev, dtype=torch.float32)
    lib = _lib.load()
    ws = torch.empty(lib.attwarp_axis_sums_workspace_bytes(B, H, W), device=dev, dtype=torch.uint8)
    with torch.cuda.device(dev):
        call("attwarp_gt_marginals", ptr(x), B, H, W, ptr(px), ptr(py), ptr(ws), stream_ptr(dev))
    return px.to(A.dtype) if A.dtype.is_floating_point else px, py.to(A.dtype) if A.dtype.is_floating_point else py


def resample_cdf(Fcdf: torch.Tensor, target_len: int) -> torch.Tensor:
    """Reference :53-62."""
    dev = require_gpu(Fcdf)
    if _lib.needs_grad(Fcdf):
        up = torch.nn.functional.interpolate(_make_strictly_increasing(Fcdf.float()).unsqueeze(1), size=int(target_len),
                                             mode="linear", align_corners=True).squeeze(1)
        return _make_strictly_increasing(up)
    F = _f32c(Fcdf)
    B, N = F.shape
    out = torch.empty(B, int(target_len), device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        call("attwarp_resample_cdf", ptr(F), B, N, int(target_len), ptr(out), stream_ptr(dev))
    return out


def _right_inverse_forward(yN: torch.Tensor, L_in: int, eps: float):
    dev = yN.device
    L_out = yN.shape[1]
    inv = _tables.right_inverse_inv(L_out, L_in, eps, dev)
    y32 = _f32c(yN)
    out = torch.empty(y32.shape[0], L_in, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        call("attwarp_upsample_pdf_right_inverse", ptr(y32), y32.shape[0], L_out, L_in, ptr(inv), ptr(out),
             stream_ptr(dev))
    return out, inv


class _RightInverse(torch.autograd.Function):
    """x_hat = A^T (A A^T + eps I)^-1 y is linear in y: forward on the HIP kernel, backward
    dy = (A A^T + eps I)^-1 (A dx) = adaptive_avg_pool1d(dx) @ inv (inv is symmetric).  The reference keeps this
    call on the autograd path of the training loss (MN/trainer.py:217-218)."""

    @staticmethod
    def forward(ctx, yN, L_in, eps):
        out, inv = _right_inverse_forward(yN, L_in, eps)
        ctx.inv, ctx.L_out, ctx.in_dtype = inv, yN.shape[1], yN.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        pooled = torch.nn.functional.adaptive_avg_pool1d(g.float().unsqueeze(1), ctx.L_out).squeeze(1)
        return (pooled.double() @ ctx.inv).to(ctx.in_dtype), None, None


def upsample_pdf_right_inverse(y: torch.Tensor, target_len: int, eps: float = 1e-8) -> torch.Tensor:
    """Reference :64-131: x_hat = A^T (A A^T + eps I)^-1 y for (L,), (B,L) or (B,C,L) inputs.  Differentiable
    with respect to ``y`` (autograd.Function around the kernel)."""
    if y.dim() not in (1, 2, 3):
        raise ValueError(f"upsample_pdf_right_inverse expects 1D/2D/3D y; got shape {tuple(y.shape)}")
    require_gpu(y)
    L_out = y.shape[-1]
    L_in = int(target_len)
    if _lib.needs_grad(y):
        out = _RightInverse.apply(y.reshape(-1, L_out), L_in, float(eps))
    else:
        out = _right_inverse_forward(y.reshape(-1, L_out), L_in, float(eps))[0]
    out = out.reshape(*y.shape[:-1], L_in)
    return out.to(y.dtype) if y.dtype.is_floating_point else out


def axis_maps_from_cdf(Fx_img: torch.Tensor, Fy_img: torch.Tensor, out_size: Tuple[int, int]
                       ) -> Tuple[torch.Tensor, torch.Tensor]:
    """The grid-construction half of ``warp_from_cdf_torch`` (:167-193) as 1-D maps:
    (map_x (B,W_out), map_y (B,H_out)) float32 source coordinates."""
    dev = require_gpu(Fx_img, Fy_img)
    Fx, Fy = _f32c(Fx_img), _f32c(Fy_img)
    B = Fx.shape[0]
    H_out, W_out = int(out_size[0]), int(out_size[1])
    mx = torch.empty(B, W_out, device=dev, dtype=torch.float32)
    my = torch.empty(B, H_out, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        st = stream_ptr(dev)
        call("attwarp_axis_map_from_cdf", ptr(Fx), B, Fx.shape[1], W_out, ptr(mx), st)
        call("attwarp_axis_map_from_cdf", ptr(Fy), B, Fy.shape[1], H_out, ptr(my), st)
    return mx, my


def remap_separable(img: torch.Tensor, map_x: torch.Tensor, map_y: torch.Tensor, mode: str = "cv2",
                    channels_last: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """cv2.remap(INTER_LINEAR, BORDER_REPLICATE) with separable maps on a batch.
    img (B,C,H,W) [or (B,H,W,C) if channels_last] uint8 / float32 / float64; maps (B,W_out), (B,H_out).
    (float64: OpenCV's CV_64F arithmetic on the generic gather kernel -- the dtype pass-through of
    ``warp_from_cdf_torch``; the roofline kernels are float32 / uint8.)"""
    dev = require_gpu(img, map_x, map_y)
    if img.dtype not in (torch.float32, torch.uint8, torch.float64):
        raise TypeError(f"remap_separable: float32, uint8 or float64 image expected, got {img.dtype}")
    x = img.detach().contiguous()
    if channels_last:
        B, H, W, C = x.shape
    else:
        B, C, H, W = x.shape
    W_out, H_out = map_x.shape[1], map_y.shape[1]
    mx, my = _f32c(map_x), _f32c(map_y)
    shape = (B, H_out, W_out, C) if channels_last else (B, C, H_out, W_out)
    if out is None:
        out = torch.empty(shape, device=dev, dtype=x.dtype)
    elif tuple(out.shape) != shape or out.dtype != x.dtype or out.device != dev or not out.is_contiguous():
        raise ValueError(f"remap_separable: out must be a contiguous {x.dtype} tensor of shape {shape} on {dev}")
    with torch.cuda.device(dev):
        call("attwarp_remap_bilinear", ptr(x), ptr(out), _lib.dtype_id(x), _lib.HWC if channels_last else _lib.CHW,
             B, C, H, W, H_out, W_out, ptr(mx), ptr(my), _lib.MODE_IDS[mode], stream_ptr(dev))
    return out


def warp_from_cdf_torch(img: torch.Tensor, Fx_img: torch.Tensor, Fy_img: torch.Tensor,
                        out_size: Optional[tuple] = None, mode: str = "cv2") -> torch.Tensor:
    """Reference :133-204.

    img (B,C,H,W) uint8 or float; Fx_img (B,W), Fy_img (B,H) CDFs in [0,1];
    out_size (H_out, W_out) or None.  Returns (B,C,H_out,W_out) on img's device and dtype.
    Raises AssertionError for a non-4-D image (:146) and ValueError when a CDF length does not
    match the image (:162-165), like the reference.
    """
    assert img.ndim == 4, f"img must be (B,C,H,W); got {img.shape}"
    B, C, H, W = img.shape
    H_out, W_out = (H, W) if out_size is None else out_size
    if B == 0:
        raise ValueError("need at least one array to stack")      # the reference's np.stack([]) (:201)
    if Fx_img.shape[-1] != W:
        raise ValueError(f"Fx_img[0] length {Fx_img.shape[-1]} != image width W={W}")
    if Fy_img.shape[-1] != H:
        raise ValueError(f"Fy_img[0] length {Fy_img.shape[-1]} != image height H={H}")
    mx, my = axis_maps_from_cdf(Fx_img.reshape(B, W), Fy_img.reshape(B, H), (H_out, W_out))
    # the reference hands the image to cv2.remap in its own dtype (:152, :195-203): uint8, float32 and float64 are
    # resampled as such; float16 / bfloat16 (which cv2.remap cannot take) go through float32 and are rounded back
    if img.dtype in (torch.uint8, torch.float32, torch.float64):
        return remap_separable(img, mx, my, mode)
    out = remap_separable(img.to(torch.float32), mx, my, mode)
    return out.to(img.dtype)
