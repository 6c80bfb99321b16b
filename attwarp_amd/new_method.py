"""Drop-in for the warp core of ``Attention Guided Warping/new_method.py`` (reference :134-283,
:355-506): numpy / PIL in -> numpy out, compute on the GPU.

The reference keeps the selected transform in module globals (``set_transform_function`` :378-403,
which makes ``save_warped_image`` non re-entrant).  The same function exists here for call-site
compatibility, but every compute entry point also takes the transform as explicit arguments and the
kernels are stateless.
"""
from __future__ import annotations

import sys
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, stream_ptr

EPSILON = 1e-9
BASE_ATTENTION = 1e-9

# module-level state, mirrors the reference's globals
ATTENTION_TRANSFORM = "sqrt"          # reference default (:191)
EXP_SCALE = 1.0
EXP_DIVISOR = 1.0
APPLY_INVERSE_TO_MARGINALS = False

_KNOWN = ("identity", "square", "sqrt", "exp", "log")


def set_transform_function(transform_name, exp_scale=1.0, exp_divisor=1.0, apply_inverse=False):
    """Reference :378-403: select the attention transform; unknown names fall back to identity."""
    global ATTENTION_TRANSFORM, EXP_SCALE, EXP_DIVISOR, APPLY_INVERSE_TO_MARGINALS
    EXP_SCALE = exp_scale
    EXP_DIVISOR = exp_divisor
    APPLY_INVERSE_TO_MARGINALS = apply_inverse
    if transform_name in _KNOWN:
        ATTENTION_TRANSFORM = transform_name
        return transform_name
    print(f"Unknown transform: {transform_name}. Using identity transform.")
    ATTENTION_TRANSFORM = "identity"
    return "identity"


def _att_to_device(att_map: np.ndarray, dev: torch.device) -> torch.Tensor:
    """uint8 and float32 are consumed as they are (their float64 conversion is exact);
    everything else goes through ``astype(np.float64)`` like the reference (:206)."""
    a = np.asarray(att_map)
    if a.dtype not in (np.uint8, np.float32, np.float64):
        a = a.astype(np.float64)
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def attention_axis_maps(att: torch.Tensor, new_width: int, new_height: int, transform: str = "identity",
                        exp_scale: float = 1.0, exp_divisor: float = 1.0, apply_inverse: bool = False
                        ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Grid construction of ``warp_image_by_attention`` (:206-265) for a batch of attention maps.
    att [B,h,w] (uint8 / float32 / float64, GPU) -> (map_x [B,new_w], map_y [B,new_h]) float32."""
    dev = _lib.require_gpu(att)
    a = att.contiguous()
    B, h, w = a.shape
    if transform not in _KNOWN:
        transform = "identity"
    mx = torch.empty(B, int(new_width), device=dev, dtype=torch.float32)
    my = torch.empty(B, int(new_height), device=dev, dtype=torch.float32)
    lib = _lib.load()
    ws = torch.empty(lib.attwarp_axis_sums_workspace_bytes(B, h, w), device=dev, dtype=torch.uint8)
    with torch.cuda.device(dev):
        call("attwarp_axis_maps_from_attention", ptr(a), _lib.dtype_id(a), B, h, w, int(new_width), int(new_height),
             _lib.TRANSFORM_IDS[transform], float(exp_scale), float(exp_divisor), int(bool(apply_inverse)), ptr(mx),
             ptr(my), ptr(ws), stream_ptr(dev))
    return mx, my


def remap_hwc(image: torch.Tensor, map_x: torch.Tensor, map_y: torch.Tensor, mode: str = "cv2") -> torch.Tensor:
    """image [B,H,W,C] uint8/float32 (GPU) -> [B,H_out,W_out,C]."""
    from .checkpoint_utils import remap_separable
    return remap_separable(image, map_x, map_y, mode=mode, channels_last=True)


def warp_image_by_attention(image: np.ndarray, att_map: np.ndarray, new_width: int, new_height: int,
                            transform: Optional[str] = None, exp_scale: Optional[float] = None,
                            exp_divisor: Optional[float] = None, apply_inverse: Optional[bool] = None,
                            mode: str = "cv2") -> np.ndarray:
    """Reference :198-283.  image [h,w,3] (or [h,w]) uint8/float32, att_map [h,w] -> warped
    [new_height,new_width,3].  Transform arguments default to the module state set by
    ``set_transform_function`` (the reference's behaviour).  ``mode="cv2"`` (default) reproduces the
    arithmetic of the reference's ``cv2.remap(INTER_LINEAR, BORDER_REPLICATE)`` call (:268-271)."""
    transform = ATTENTION_TRANSFORM if transform is None else transform
    exp_scale = EXP_SCALE if exp_scale is None else exp_scale
    exp_divisor = EXP_DIVISOR if exp_divisor is None else exp_divisor
    apply_inverse = APPLY_INVERSE_TO_MARGINALS if apply_inverse is None else apply_inverse
    dev = _lib.default_device()
    img = np.asarray(image)
    if img.dtype not in (np.uint8, np.float32):
        raise TypeError(f"warp_image_by_attention: uint8 or float32 image expected, got {img.dtype}")
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    att = _att_to_device(att_map, dev)
    if att.dim() != 2 or tuple(att.shape) != img.shape[:2]:
        raise ValueError(f"attention map {tuple(att.shape)} must match image {img.shape[:2]}")
    mx, my = attention_axis_maps(att.unsqueeze(0), new_width, new_height, transform, exp_scale, exp_divisor,
                                 apply_inverse)
    out = remap_hwc(torch.from_numpy(np.ascontiguousarray(img)).to(dev).unsqueeze(0), mx, my, mode)[0]
    res = out.cpu().numpy()
    return res[:, :, 0] if squeeze else res


def resize_image_to_match_attmap(image: Optional[np.ndarray], att_map: Optional[np.ndarray], mode: str = "cv2"):
    """Reference :355-376.  Same size -> copy.  Otherwise ``cv2.resize(image, (w, h), interpolation=INTER_LINEAR)`` to
    the attention map's size.  ``mode="cv2"`` (default): OpenCV's published arithmetic -- half-pixel centres, 11-bit
    fixed-point coefficients for uint8, an exact 2 x 2 decimation as INTER_AREA (``attwarp_resize_linear``; parity
    unpinned like the resample, OpenCV is absent here); ``mode="exact"``: unquantised bilinear at the same centres.
    uint8 and float32 images; other dtypes raise."""
    if image is None or att_map is None:
        return None
    th, tw = att_map.shape[:2]
    ch, cw = image.shape[:2]
    if (ch, cw) == (th, tw):
        return image.copy()
    dev = _lib.default_device()
    img = np.asarray(image)
    if img.dtype not in (np.uint8, np.float32):
        raise TypeError(f"resize_image_to_match_attmap: uint8 or float32 image expected, got {img.dtype}")
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    src = torch.from_numpy(np.ascontiguousarray(img)).to(dev).unsqueeze(0)
    if mode == "cv2":
        out_t = torch.empty((1, th, tw, img.shape[2]), device=dev, dtype=src.dtype)
        with torch.cuda.device(dev):
            _lib.call("attwarp_resize_linear", _lib.ptr(src), _lib.ptr(out_t), _lib.dtype_id(src), 1, img.shape[2], ch, cw,
                      th, tw, _lib.stream_ptr(dev))
        out = out_t[0].cpu().numpy()
    elif mode == "exact":
        mx = ((torch.arange(tw, dtype=torch.float64) + 0.5) * (cw / tw) - 0.5).clamp(0, cw - 1).float().to(dev)[None]
        my = ((torch.arange(th, dtype=torch.float64) + 0.5) * (ch / th) - 0.5).clamp(0, ch - 1).float().to(dev)[None]
        out = remap_hwc(src, mx, my, "exact")[0].cpu().numpy()
    else:
        raise ValueError(f"resize_image_to_match_attmap: unknown mode {mode!r}")
    return out[:, :, 0] if squeeze else out


def _coerce_att_map(att_map, width, height):
    """Reference :431-452."""
    from PIL import Image
    if isinstance(att_map, np.ndarray):
        pass
    elif isinstance(att_map, Image.Image):
        att_map = np.array(att_map)
    elif isinstance(att_map, list):
        if len(att_map) > 0:
            att_map = np.array(att_map[0])
        else:
            att_map = np.ones((height, width), dtype=np.float32) * 128
    if att_map.ndim == 3:
        att_map = np.mean(att_map, axis=2)
    elif att_map.ndim != 2:
        raise ValueError(f"Attention map must be 2D, got shape {att_map.shape}")
    return att_map


def save_warped_image(image_path, att_map, original_image_save_path, masked_overlay_save_path, output_path,
                      vis_path=None, width=500, height=500, transform="identity", exp_scale=1.0, exp_divisor=1.0,
                      apply_inverse=False, attention_alpha=0.5, mode="cv2"):
    """Reference :405-506.  Warps ``image_path`` (path or PIL image) by ``att_map`` and writes the
    result to ``output_path``; returns True, or prints the error and returns False (the reference
    swallows every exception, :504-506).  Images are handled in BGR order like the reference;
    files are written with Pillow.  The overlay / visualisation strip outputs are out of scope
    (visualisation only): ``masked_overlay_save_path`` and ``vis_path`` are accepted and ignored.
    ``mode`` (not in the reference) selects the resample arithmetic: "cv2" (default, what the reference's
    ``cv2.remap`` call computes) or "exact"."""
    try:
        from PIL import Image
        if isinstance(image_path, str):
            try:
                image = np.array(Image.open(image_path).convert("RGB"))[:, :, ::-1]
            except Exception:
                raise ValueError(f"Could not read image: {image_path}")
        else:
            image = _cvt_rgb2bgr(np.array(image_path))
        image = np.ascontiguousarray(image)
        if original_image_save_path:
            _imwrite(original_image_save_path, image)
        att_map = _coerce_att_map(att_map, width, height)
        image_for_warping = resize_image_to_match_attmap(image, att_map, mode)
        if image_for_warping is None:
            raise ValueError("Failed to resize image to match attention map dimensions for warping")
        name = set_transform_function(transform, exp_scale, exp_divisor, apply_inverse)
        warped = warp_image_by_attention(image_for_warping, att_map, width, height, name, exp_scale, exp_divisor,
                                         apply_inverse, mode)
        if warped is None:
            raise ValueError("Warping failed")
        _imwrite(output_path, warped)
        return True
    except Exception as e:  # noqa: BLE001 - reference behaviour
        print(f"Error during processing: {e}", file=sys.stderr)
        return False


def _cvt_rgb2bgr(image: np.ndarray) -> np.ndarray:
    """``cv2.cvtColor(image, cv2.COLOR_RGB2BGR)`` as the reference applies it to ``np.array(PIL image)`` (:421-422), with
    OpenCV's input checks: the conversion takes 3- or 4-channel sources of depth uint8 / uint16 / float32 (OpenCV
    ``cvtColorBGR2BGR``: ``CvtHelper<Set<3, 4>, Set<3, 4>, Set<CV_8U, CV_16U, CV_32F>>``, destination 3 channels) and raises
    ``cv2.error`` for everything else -- a mode-"L" / "1" / "I" / "F" image arrives as a 2-D array and FAILS there, so the
    reference's ``save_warped_image`` prints the error and returns False (:504-506); an RGBA image loses its alpha channel.
    The same conditions raise here (ValueError standing in for cv2.error)."""
    if image.ndim != 3 or image.shape[2] not in (3, 4):
        raise ValueError(f"cvtColor(COLOR_RGB2BGR): invalid number of channels in input image: shape {image.shape} "
                         "(3 or 4 channels expected)")
    if image.dtype not in (np.uint8, np.uint16, np.float32):
        raise ValueError(f"cvtColor(COLOR_RGB2BGR): unsupported depth of input image: {image.dtype}")
    return image[:, :, 2::-1]                          # R,G,B[,A] -> B,G,R


def _imwrite(path, bgr: np.ndarray):
    from PIL import Image
    arr = bgr[:, :, ::-1] if bgr.ndim == 3 and bgr.shape[2] == 3 else bgr
    Image.fromarray(np.ascontiguousarray(arr)).save(path)
