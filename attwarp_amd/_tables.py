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


def right_inverse_inv(n_out: int, n_in: int, eps: float, device: torch.device) -> torch.Tensor:
    key = ("inv", n_out, n_in, float(eps), str(device))
    t = _DEV_CACHE.get(key)
    if t is None:
        t = torch.from_numpy(_right_inverse_inv_host(n_out, n_in, float(eps))).to(device)
        _DEV_CACHE[key] = t
    return t


_PRECISION_BITS = 32 - 8 - 2


def _sinc(x: float) -> float:
    if x == 0.0:
        return 1.0
    x *= math.pi
    return math.sin(x) / x


def _lanczos3(x: float) -> float:
    return _sinc(x) * _sinc(x / 3.0) if -3.0 <= x < 3.0 else 0.0


def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


_FILTERS = {"lanczos": (_lanczos3, 3.0), "bicubic": (_bicubic, 2.0)}


@functools.lru_cache(maxsize=64)
def _lanczos_tables_host(n_in: int, n_out: int, filt: str = "lanczos"):
    fn, support0 = _FILTERS[filt]
    if n_in == n_out:       # Pillow skips the pass; an identity table makes clip8((v << 22 + 2^21) >> 22) == v
        bounds = np.stack([np.arange(n_out, dtype=np.int32), np.ones(n_out, dtype=np.int32)], axis=1)
        return bounds, np.full((n_out, 1), 1 << _PRECISION_BITS, dtype=np.int32), 1
    scale = n_in / n_out
    fscale = max(scale, 1.0)
    support = support0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((n_out, 2), dtype=np.int32)
    kk = np.zeros((n_out, ksize), dtype=np.int32)
    inv_fscale = 1.0 / fscale
    for o in range(n_out):
        center = (o + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        hi = min(int(center + support + 0.5), n_in)
        cnt = hi - lo
        w = [fn((i + lo - center + 0.5) * inv_fscale) for i in range(cnt)]
        tot = 0.0
        for v in w:
            tot += v
        if tot != 0.0:
            w = [v / tot for v in w]
        for i, v in enumerate(w):
            kk[o, i] = int((-0.5 if v < 0 else 0.5) + v * (1 << _PRECISION_BITS))
        bounds[o] = (lo, cnt)
    return bounds, kk, ksize


def lanczos_tables(n_in: int, n_out: int, device: torch.device, filt: str = "lanczos"):
    """Pillow 8-bit coefficient tables on the device: (bounds int32 [n_out,2], kk int32 [n_out,ksize], ksize)."""
    key = (filt, n_in, n_out, str(device))
    t = _DEV_CACHE.get(key)
    if t is None:
        bounds, kk, ksize = _lanczos_tables_host(n_in, n_out, filt)
        if ksize < 8:           # rows of exactly 8 zero-padded coefficients enable the register-window kernels (attwarp.h)
            kk = np.concatenate([kk, np.zeros((kk.shape[0], 8 - ksize), dtype=np.int32)], axis=1)
            ksize = 8
        t = (torch.from_numpy(bounds).to(device), torch.from_numpy(kk).to(device), ksize)
        _DEV_CACHE[key] = t
    return t
