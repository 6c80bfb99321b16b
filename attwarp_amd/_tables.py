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


def right_inverse_inv(n_out: int, n_in: int, eps: float, device: torch.device) -> torch.Tensor:
    key = ("inv", n_out, n_in, float(eps), str(device))
    t = _DEV_CACHE.get(key)
    if t is None:
        t = torch.from_numpy(_right_inverse_inv_host(n_out, n_in, float(eps))).to(device)
        _DEV_CACHE[key] = t
    return t


_FILTER_IDS = {"lanczos": 0, "bicubic": 1}          # ATTWARP_PIL_LANCZOS / ATTWARP_PIL_BICUBIC


@functools.lru_cache(maxsize=8192)
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
        while len(_DEV_CACHE) >= _DEV_CACHE_MAX:          # bounded: ragged batches bring two tables per distinct image size
            _DEV_CACHE.pop(next(iter(_DEV_CACHE)))
        bounds, kk, ksize = _lanczos_tables_host(n_in, n_out, filt)
        if ksize < 8:           # rows of exactly 8 zero-padded coefficients enable the register-window kernels (attwarp.h)
            kk = np.concatenate([kk, np.zeros((kk.shape[0], 8 - ksize), dtype=np.int32)], axis=1)
            ksize = 8
        t = (torch.from_numpy(bounds).to(device), torch.from_numpy(kk).to(device), ksize)
        _DEV_CACHE[key] = t
    return t
