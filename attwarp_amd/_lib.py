"""ctypes binding of libattwarp_hip.so (C ABI declared in include/attwarp.h).

The HIP library is the product: there is NO CPU fallback.  Importing this module
on a machine without the built library raises ImportError; calling any op on a
non-GPU tensor raises.  Build with ``python __graft_entry__.py`` or
``make -C attwarp_amd/csrc``.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
from ctypes import c_double, c_float, c_int, c_int64, c_size_t, c_void_p, c_char_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libattwarp_hip.so")
# the same sources built with -DATTWARP_TUNING: attwarp_debug_set() can force kernel variants.  Loaded only by
# debug_override() (parity tests, A/B tools); the product never touches it.
TUNING_LIB_PATH = os.path.join(_HERE, "csrc", "libattwarp_hip_tuning.so")

# enums of include/attwarp.h
F32, F16, BF16, U8, F64 = 0, 1, 2, 3, 4
HWC, CHW = 0, 1
EXACT, CV2 = 0, 1
TRANSFORM_IDS = {"identity": 0, "square": 1, "sqrt": 2, "exp": 3, "log": 4}
MODE_IDS = {"exact": EXACT, "cv2": CV2}

_DTYPE_IDS = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16, torch.uint8: U8, torch.float64: F64}

# name -> (restype, argtypes); must list every symbol the header declares
SIGNATURES = {
    "attwarp_version": (c_int, []),
    "attwarp_last_error": (c_char_p, []),
    "attwarp_attn_reduce_step": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64,
                                          c_int64, c_void_p, c_int, c_void_p, c_void_p]),
    "attwarp_attn_finalize": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "attwarp_attn_reduce_stack_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "attwarp_attn_reduce_stack": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p,
                                           c_void_p, c_void_p]),
    "attwarp_attn_probe_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "attwarp_attn_probe_last_query": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int64,
                                               c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int, c_float,
                                               c_void_p, c_void_p, c_void_p]),
    "attwarp_masked_token_mean": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "attwarp_film_axis_means": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "attwarp_mask_postproc": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "attwarp_mask_upsample_lanczos": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                               c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                               c_void_p]),
    "attwarp_clip_preprocess_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                            c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_int, c_void_p]),
    "attwarp_adaptive_avg_pool": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "attwarp_axis_sums_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "attwarp_gt_marginals": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "attwarp_safe_softmax": (c_int, [c_void_p, c_int, c_int, c_float, c_void_p, c_void_p]),
    "attwarp_upsample_pdf_right_inverse": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "attwarp_cdf_from_density": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "attwarp_make_strictly_increasing": (c_int, [c_void_p, c_int, c_int, c_double, c_void_p, c_void_p]),
    "attwarp_resample_cdf": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "attwarp_axis_map_from_cdf": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "attwarp_axis_maps_from_pdf": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                            c_void_p, c_void_p, c_void_p, c_void_p]),
    "attwarp_axis_maps_from_steps": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "attwarp_axis_maps_from_attention": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                                  c_double, c_double, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "attwarp_warp_step_fused": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                         c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "attwarp_warp_step_fused_slots": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                               c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "attwarp_attn_reduce_and_maps": (c_int, [c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                              c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p]),
    "attwarp_mask_chain_step": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                         c_void_p, c_int, c_int, c_float, c_void_p,
                                         c_int, c_double, c_double, c_int, c_void_p, c_void_p]),
    "attwarp_attention_transform_lut": (c_int, [c_int, c_double, c_double, c_void_p, c_void_p]),
    "attwarp_pil_coeffs_8bpc": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_int]),
    "attwarp_ragged_table_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int, c_int, c_int]),
    "attwarp_ragged_plan": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t]),
    "attwarp_mask_chain_ragged": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_int, c_int, c_int, c_float, c_void_p,
                                           c_int, c_double, c_double, c_int, c_void_p, c_void_p]),
    "attwarp_axis_maps_from_steps_t": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "attwarp_resize_linear": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "attwarp_remap_bilinear": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                        c_void_p, c_void_p, c_int, c_void_p]),
}

PIL_LANCZOS, PIL_BICUBIC = 0, 1


class RaggedImage(ctypes.Structure):
    """attwarp_ragged_image (include/attwarp.h): one image of a ragged batch, device pointers in a host array."""
    _fields_ = [("image", c_void_p), ("bounds_x", c_void_p), ("kk_x", c_void_p), ("bounds_y", c_void_p), ("kk_y", c_void_p),
                ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("ksize_x", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class RaggedHeader(ctypes.Structure):
    """attwarp_ragged_header: the first bytes of a table written by attwarp_ragged_plan."""
    _fields_ = [("magic", ctypes.c_uint32), ("B", ctypes.c_int32), ("C", ctypes.c_int32), ("g", ctypes.c_int32),
                ("H_out", ctypes.c_int32), ("W_out", ctypes.c_int32),
                ("nL", ctypes.c_int32), ("nP", ctypes.c_int32), ("nR", ctypes.c_int32), ("nplans", ctypes.c_int32),
                ("rows_per_block", ctypes.c_int32), ("blocks_per_image", ctypes.c_int32), ("kd", ctypes.c_int32),
                ("max_hw", ctypes.c_int32),
                ("table_bytes", ctypes.c_uint64), ("mota_bytes", ctypes.c_uint64), ("sums_bytes", ctypes.c_uint64),
                ("lds_bytes", ctypes.c_uint64), ("off_images", ctypes.c_uint64), ("off_plans", ctypes.c_uint64),
                ("off_lmap", ctypes.c_uint64), ("off_pmap", ctypes.c_uint64), ("off_order", ctypes.c_uint64)]


_lib = None            # the library every call goes to: the product, or the tuning flavour inside debug_override()
_product = None
_tuning = None


class AttWarpError(RuntimeError):
    """A libattwarp_hip.so entry point returned a negative status."""


def _open(path):
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError = header / library mismatch
        fn.restype = res
        fn.argtypes = args
    return lib


def load():
    """Load (once) and return the ctypes handle.  Fails loudly if the library is missing."""
    global _lib, _product
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"attwarp_amd: HIP library not built ({LIB_PATH} missing). Run `python __graft_entry__.py` or "
            f"`make -C attwarp_amd/csrc`. There is no CPU fallback.")
    _product = _open(LIB_PATH)
    _lib = _product
    return _lib


def load_tuning():
    """The tuning flavour (test / measurement only): same kernels, plus attwarp_debug_set()."""
    global _tuning
    if _tuning is None:
        if not os.path.exists(TUNING_LIB_PATH):
            raise ImportError(f"attwarp_amd: tuning library not built ({TUNING_LIB_PATH} missing); run `make -C attwarp_amd/csrc`")
        _tuning = _open(TUNING_LIB_PATH)
        _tuning.attwarp_debug_set.restype = c_int
        _tuning.attwarp_debug_set.argtypes = [c_char_p, c_int, ctypes.POINTER(c_int)]
        _tuning.attwarp_debug_stream_copy.restype = c_int
        _tuning.attwarp_debug_stream_copy.argtypes = [c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]
    return _tuning


def call(name: str, *args):
    """Invoke an int-returning entry point; raise AttWarpError on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.attwarp_last_error().decode("utf-8", "replace")
        raise AttWarpError(f"{name} failed ({rc}): {msg}")


@contextlib.contextmanager
def debug_override(**settings: int):
    """Test / measurement hook: run the calls inside the context on the TUNING flavour of the library with kernel
    variants forced through ``attwarp_debug_set`` -- e.g. ``with debug_override(remap_variant=1): ...`` runs the generic
    gather kernel.  On exit every key gets back the value it had before (nesting works) and calls go to the product
    library again.  Process wide and single threaded: do not use it while other threads launch kernels."""
    global _lib
    load()
    tl = load_tuning()
    outer, _lib = _lib, tl
    prev = {}
    try:
        for k, v in settings.items():
            old = c_int(-1)
            rc = tl.attwarp_debug_set(k.encode(), int(v), ctypes.byref(old))
            if rc != 0:
                raise AttWarpError(f"attwarp_debug_set failed ({rc}): {tl.attwarp_last_error().decode('utf-8', 'replace')}")
            prev[k] = old.value
        yield
    finally:
        for k, v in prev.items():
            tl.attwarp_debug_set(k.encode(), int(v), None)
        _lib = outer


def ptr(t: torch.Tensor | None):
    return None if t is None else c_void_p(t.data_ptr())


def dtype_id(t: torch.Tensor) -> int:
    try:
        return _DTYPE_IDS[t.dtype]
    except KeyError:
        raise TypeError(f"attwarp_amd: unsupported dtype {t.dtype}") from None


def require_gpu(*tensors: torch.Tensor) -> torch.device:
    """All tensors must live on the same GPU; returns it.  No CPU path exists."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "attwarp_amd: tensor is on %s; the HIP kernels need a GPU tensor (there is no CPU fallback)" % t.device)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"attwarp_amd: tensors on different devices ({dev} vs {t.device})")
    if dev is None:
        raise RuntimeError("attwarp_amd: no tensor given")
    return dev


def needs_grad(*tensors) -> bool:
    """True when autograd has to see through the call (grad mode on and some input requires grad).  The HIP kernels
    are forward-only; the training-time call sites of the reference (``net(...)`` and
    ``upsample_pdf_right_inverse(px_s, ...)`` on the loss path, MN/trainer.py:210-260) then take a differentiable
    route: stock PyTorch-ROCm ops on the same GPU, or an autograd.Function around the kernel."""
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


def stream_ptr(dev: torch.device):
    """The caller's current HIP stream on `dev` (kernels are stream ordered, never synchronise)."""
    return c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def default_device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("attwarp_amd: no GPU visible; the numpy-facing entry points run their kernels on cuda:0 "
                           "(there is no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())
