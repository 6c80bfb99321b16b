"""ctypes loader for oracle/liboracle_warp.so (the plain-C restatement, TEST INFRASTRUCTURE ONLY --
same import rules as warp_oracle.py)."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_int, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(_HERE, "liboracle_warp.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(PATH):
            raise ImportError(f"{PATH} missing: run `make -C oracle` (or __graft_entry__.build())")
        _lib = ctypes.CDLL(PATH)
    return _lib


def _p(a):
    return a.ctypes.data_as(c_void_p)


def attn_reduce_stack(rows, starts, ntok=576):
    rows = np.ascontiguousarray(rows, np.float32)
    T, B, heads, kv = rows.shape
    st = np.ascontiguousarray(starts, np.int32)
    out = np.empty((B, ntok), np.float32)
    load().oracle_attn_reduce_stack_f32(_p(rows), c_int(T), c_int(B), c_int(heads), c_int(kv), _p(st), c_int(ntok), _p(out))
    return out


def marginals(A):
    A = np.ascontiguousarray(A, np.float32)
    H, W = A.shape
    px = np.empty(W, np.float32); py = np.empty(H, np.float32)
    load().oracle_marginals(_p(A), c_int(H), c_int(W), _p(px), _p(py))
    return px, py


def right_inverse(y, L, inv, clamp0=False):
    y = np.ascontiguousarray(y, np.float32); inv = np.ascontiguousarray(inv, np.float64)
    out = np.empty(L, np.float32)
    load().oracle_right_inverse(_p(y), c_int(y.shape[0]), c_int(L), _p(inv), c_int(int(clamp0)), _p(out))
    return out


def cdf_from_density(p):
    p = np.ascontiguousarray(p, np.float32)
    F = np.empty_like(p)
    load().oracle_cdf_from_density(_p(p), c_int(p.shape[0]), _p(F))
    return F


def axis_map_from_cdf(F, n_out):
    F = np.ascontiguousarray(F, np.float32)
    m = np.empty(n_out, np.float32)
    load().oracle_axis_map_from_cdf(_p(F), c_int(F.shape[0]), c_int(n_out), _p(m))
    return m


def remap_bilinear(src, mx, my, layout="hwc", mode="cv2"):
    src = np.ascontiguousarray(src, np.float32)
    mx = np.ascontiguousarray(mx, np.float32); my = np.ascontiguousarray(my, np.float32)
    if layout == "hwc":
        H, W, C = src.shape; out = np.empty((my.shape[0], mx.shape[0], C), np.float32); lid = 0
    else:
        C, H, W = src.shape; out = np.empty((C, my.shape[0], mx.shape[0]), np.float32); lid = 1
    load().oracle_remap_bilinear_f32(_p(src), _p(out), c_int(lid), c_int(C), c_int(H), c_int(W), c_int(my.shape[0]),
                                     c_int(mx.shape[0]), _p(mx), _p(my), c_int(int(mode == "cv2")))
    return out


def remap_bilinear_u8(src, mx, my, mode="cv2"):
    """uint8 [H,W,C] -> [H_out,W_out,C] (the arithmetic of warp_oracle.remap_bilinear on uint8 sources)."""
    src = np.ascontiguousarray(src, np.uint8)
    mx = np.ascontiguousarray(mx, np.float32); my = np.ascontiguousarray(my, np.float32)
    H, W, C = src.shape
    out = np.empty((my.shape[0], mx.shape[0], C), np.uint8)
    load().oracle_remap_bilinear_u8(_p(src), _p(out), c_int(C), c_int(H), c_int(W), c_int(my.shape[0]), c_int(mx.shape[0]),
                                    _p(mx), _p(my), c_int(int(mode == "cv2")))
    return out


def warp_from_attention_stack(img, rows, start, inv_x, inv_y, layout="hwc", mode="cv2"):
    """One image through the whole path.  rows [T,heads,kv]."""
    img = np.ascontiguousarray(img, np.float32); rows = np.ascontiguousarray(rows, np.float32)
    if layout == "hwc":
        H, W, C = img.shape; lid = 0
    else:
        C, H, W = img.shape; lid = 1
    out = np.empty_like(img)
    T, heads, kv = rows.shape
    load().oracle_warp_from_attention_stack_f32(_p(img), _p(out), c_int(lid), c_int(C), c_int(H), c_int(W), _p(rows),
                                                c_int(T), c_int(heads), c_int(kv), c_int(int(start)),
                                                _p(np.ascontiguousarray(inv_x, np.float64)),
                                                _p(np.ascontiguousarray(inv_y, np.float64)), c_int(int(mode == "cv2")))
    return out
