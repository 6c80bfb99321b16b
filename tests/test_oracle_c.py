"""The plain-C restatement (oracle/warp_ref.c, used for bench.py's cpu_baseline) must agree
bit-for-bit with the numpy oracle that is pinned to the reference goldens.  CPU only."""
import subprocess, os
import numpy as np
import pytest

from oracle import warp_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def C():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    from oracle import c_oracle
    c_oracle.load()
    return c_oracle


def test_c_attn_reduce(golden, C):
    g = golden("attn_reduce")
    rows = np.stack([g[f"step_in_{t}"][:, :, -1, :] for t in range(4)])
    assert np.array_equal(C.attn_reduce_stack(rows, g["starts"]), O.attn_reduce_stack(rows, g["starts"]))


def test_c_axis_chain(golden, C):
    g = golden("pdf_cdf")
    y = g["y"]
    for L in (336, 500, 1024):
        inv = O.right_inverse_core(24, L)
        x_o = O.upsample_pdf_right_inverse(y, L)
        for b in range(y.shape[0]):
            x_c = C.right_inverse(y[b], L, inv)
            assert np.array_equal(x_c, x_o[b])
            p = np.maximum(x_o[b], 0)
            F_o = O.cdf_from_density(p[None])[0]
            assert np.array_equal(C.cdf_from_density(p), F_o)
            assert np.array_equal(C.axis_map_from_cdf(F_o, L), O.axis_map_from_cdf(F_o, L))
            assert np.array_equal(C.axis_map_from_cdf(F_o, 500), O.axis_map_from_cdf(F_o, 500))


def test_c_map_ties(golden, C):
    g = golden("maps_from_cdf")
    for b in range(2):
        assert np.array_equal(C.axis_map_from_cdf(g["ties_F"][b], 96), g["ties_mx"][b])
        assert np.array_equal(C.axis_map_from_cdf(g["ties_F"][b], 80), g["ties_my"][b])


def test_c_marginals(C):
    rng = np.random.default_rng(3)
    A = rng.standard_normal((24, 24)).astype(np.float32)
    px, py = C.marginals(A)
    pxo, pyo = O.gt_marginals(A[None, None])
    assert np.array_equal(px, pxo[0]) and np.array_equal(py, pyo[0])


@pytest.mark.parametrize("layout", ["hwc", "chw"])
@pytest.mark.parametrize("mode", ["exact", "cv2"])
def test_c_remap(C, layout, mode):
    rng = np.random.default_rng(4)
    img = rng.random((45, 61, 3), dtype=np.float32)
    mx = np.sort(rng.random(70).astype(np.float32) * 63 - 1)
    my = np.sort(rng.random(50).astype(np.float32) * 47 - 1)
    mx[:4] = [-3.0, 0.015625, 0.046875, 2.5]          # outside, cvRound ties
    ref = O.remap_bilinear(img, mx, my, mode)
    src = img if layout == "hwc" else np.ascontiguousarray(img.transpose(2, 0, 1))
    out = C.remap_bilinear(src, mx, my, layout, mode)
    if layout == "chw":
        out = out.transpose(1, 2, 0)
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("mode", ["exact", "cv2"])
def test_c_whole_path(C, mode):
    rng = np.random.default_rng(5)
    T, heads, kv, S = 3, 32, 640, 48
    lg = rng.standard_normal((T, 1, heads, kv)).astype(np.float32)
    rows = np.exp(lg - lg.max(-1, keepdims=True)); rows = (rows / rows.sum(-1, keepdims=True)).astype(np.float32)
    img = rng.random((S, S, 3), dtype=np.float32)
    inv = O.right_inverse_core(24, S)
    out = C.warp_from_attention_stack(img, rows[:, 0], 37, inv, inv, mode=mode)
    att = O.attn_reduce_stack(rows, [37]).reshape(1, 1, 24, 24)
    px, py = O.gt_marginals(att)
    Fx = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(px, S), 0))
    Fy = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(py, S), 0))
    mx, my = O.maps_from_cdf(Fx, Fy)
    assert np.array_equal(out, O.remap_bilinear(img, mx[0], my[0], mode))
