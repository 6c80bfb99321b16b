"""Small host-side constant tables the kernels consume (computed once, cached per device).

* ``right_inverse_inv``: (A A^T + eps I)^-1 for the adaptive-avg-pool1d matrix A -- the 24x24
  system ``upsample_pdf_right_inverse`` solves on every call in the reference
  (MN/checkpoint_utils.py:104-120, incl. a Python loop with 48 ``.item()`` host syncs).
* ``lanczos_tables``: Pillow's 8-bit LANCZOS coefficient tables (``precompute_coeffs`` +
  ``normalize_coeffs_8bpc`` in Pillow's Resample.c) behind ``invtrans`` (llava.py:195-196).
"""
from __future__ import annotations

import functools
import math

import numpy as np
import torch


def _windows(n_in: int, n_out: int):
    i = np.arange(n_out, dtype=np.int64)
    return (i * n_in) // n_out, ((i + 1) * n_in + n_out - 1) // n_out


@functools.lru_cache(maxsize=64)
def _right_inverse_inv_host(n_out: int, n_in: int, eps: float) -> np.ndarray:
    starts, ends = _windows(n_in, n_out)
    A = np.zeros((n_out, n_in), dtype=np.float32)
    for k in range(n_out):
        s, e = int(starts[k]), int(ends[k])
        A[k, s:e] = np.float32(1.0) / np.float32(max(e - s, 1))
    A64 = A.astype(np.float64)
    gram = (A64[:, None, :] * A64[None, :, :]).sum(axis=2).astype(np.float32)   # reference: float32 A @ A.T
    if eps > 0:
        gram = (gram + np.float32(eps) * np.eye(n_out, dtype=np.float32)).astype(np.float32)
    return np.linalg.inv(gram.astype(np.float64))


_DEV_CACHE: dict = {}
_DEV_CACHE_MAX = 8192          # entries (callers keep the tensors they use alive; eviction only drops the cache's reference)
_DEV_CACHE_LOCK = __import__("threading").Lock()     # host threads on their own streams share the cache (insert + evict)


def _cache_put(key, value):
    """Insert under the lock, evicting the oldest entries at capacity (two threads evicting at once must not trip over each
    other's pop); returns the entry that is in the cache afterwards (another thread may have been first)."""
    with _DEV_CACHE_LOCK:
        have = _DEV_CACHE.get(key)
        if have is not None:
            return have
        while len(_DEV_CACHE) >= _DEV_CACHE_MAX:
            _DEV_CACHE.pop(next(iter(_DEV_CACHE)), None)
        _DEV_CACHE[key] = value
        return value


def right_inverse_inv(n_out: int, n_in: int, eps: float, device: torch.device) -> torch.Tensor:
    key = ("inv", n_out, n_in, float(eps), str(device))
    t = _DEV_CACHE.get(key)
    if t is None:
        t = _cache_put(key, torch.from_numpy(_right_inverse_inv_host(n_out, n_in, float(eps))).to(device))
    return t


def attention_transform_lut(transform: str, exp_scale: float, exp_divisor: float, device: torch.device):
    """The 256 float64 values ``transform(max(v, 0)) + 1e-9`` for v = 0 .. 255 (AGW/new_method.py:134-179,208-215) the
    one-launch chain steps read for the sqrt / exp / log transforms -- written once per (transform, exp_scale, exp_divisor,
    device) by ``attwarp_attention_transform_lut`` with the device functions of the stand-alone profile kernel (bit-identical)
    -- or None for identity / square (computed in registers).  Must be called OUTSIDE a stream capture the first time."""
    from . import _lib
    if transform not in ("sqrt", "exp", "log"):
        return None
    key = ("xf_lut", transform, float(exp_scale), float(exp_divisor), str(device))
    t = _DEV_CACHE.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("attwarp_amd: the transform table of a one-launch chain step must exist before a stream capture "
                               "(construct the MaskChainStream / run one eager step first)")
        lut = torch.empty(256, device=device, dtype=torch.float64)
        with torch.cuda.device(device):
            _lib.call("attwarp_attention_transform_lut", _lib.TRANSFORM_IDS[transform], float(exp_scale), float(exp_divisor),
                      _lib.ptr(lut), _lib.stream_ptr(device))
            torch.cuda.current_stream(device).synchronize()      # other streams may read it from now on
        t = _cache_put(key, lut)
    return t


_FILTER_IDS = {"lanczos": 0, "bicubic": 1}          # ATTWARP_PIL_LANCZOS / ATTWARP_PIL_BICUBIC


@functools.lru_cache(maxsize=1024)          # (host copies: ~12 bytes x taps per output pixel; the device cache below is the hot one)
def _lanczos_tables_host(n_in: int, n_out: int, filt: str = "lanczos"):
    """Pillow's 8-bit coefficient tables (precompute_coeffs + normalize_coeffs_8bpc, libImaging/Resample.c) from the
    library's HOST helper attwarp_pil_coeffs_8bpc (double arithmetic with libm's sin, as Pillow's C code; a batch of
    differently sized images needs two tables per distinct size -- a Python loop cost 6-9 ms per table).
    -> (bounds int32 [n_out,2], kk int32 [n_out,ksize], ksize)."""
    from . import _lib
    if n_in == n_out:
        cols = 1
    else:
        support = (3.0 if filt == "lanczos" else 2.0) * max(n_in / n_out, 1.0)
        cols = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((n_out, 2), dtype=np.int32)
    kk = np.zeros((n_out, cols), dtype=np.int32)
    lib = _lib.load()
    ksize = lib.attwarp_pil_coeffs_8bpc(n_in, n_out, _FILTER_IDS[filt], bounds.ctypes.data, kk.ctypes.data, cols)
    if ksize < 0:
        raise _lib.AttWarpError(f"attwarp_pil_coeffs_8bpc failed ({ksize}): {lib.attwarp_last_error().decode('utf-8', 'replace')}")
    return bounds, kk, ksize


def lanczos_tables(n_in: int, n_out: int, device: torch.device, filt: str = "lanczos"):
    """Pillow 8-bit coefficient tables on the device: (bounds int32 [n_out,2], kk int32 [n_out,ksize], ksize)."""
    key = (filt, n_in, n_out, str(device))
    t = _DEV_CACHE.get(key)
    if t is None:
        bounds, kk, ksize = _lanczos_tables_host(n_in, n_out, filt)   # (bounded cache: ragged batches bring two tables per distinct size)
        if ksize < 8:           # rows of exactly 8 zero-padded coefficients enable the register-window kernels (attwarp.h)
            kk = np.concatenate([kk, np.zeros((kk.shape[0], 8 - ksize), dtype=np.int32)], axis=1)
            ksize = 8
        t = _cache_put(key, (torch.from_numpy(bounds).to(device), torch.from_numpy(kk).to(device), ksize))
    return t
