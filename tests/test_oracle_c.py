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
@pytest.mark.parametrize("chan", [1, 3, 4])
def test_c_remap_uint8(C, mode, chan):
    """The C uint8 resample (the whole-batch checker of the GPU tests) == the numpy oracle: monotone, wild, out-of-range
    and non-finite coordinates, cvRound ties, extreme pixel values."""
    rng = np.random.default_rng(40 + chan)
    img = rng.integers(0, 256, (37, 53, chan), dtype=np.uint8)
    img[:3] = 255; img[3:5] = 0
    for kind in ("sorted", "wild"):
        mx = rng.random(64).astype(np.float32) * 57 - 2
        my = rng.random(41).astype(np.float32) * 41 - 2
        if kind == "sorted":
            mx, my = np.sort(mx), np.sort(my)
        mx[:6] = [-3.0, 0.015625, 0.046875, 2.5, np.nan, 1e30]
        my[:3] = [0.484375, 36.0, np.inf]
        with np.errstate(all="ignore"):
            ref = O.remap_bilinear(img, mx, my, mode)
        assert np.array_equal(C.remap_bilinear_u8(img, mx, my, mode), ref), kind


@pytest.mark.parametrize("mode", ["exact", "cv2"])
def test_non_finite_and_huge_coordinates(C, mode):
    """Coordinates no image has: the conventions both oracles (and the kernels) follow.  cv2: cvRound as OpenCV's x86
    builds compute it -- NaN, +-Inf and products outside int32 give INT_MIN, i.e. pixel 0 with a zero fraction, for
    either sign; exact: the coordinate is clamped to [-1, size] first, NaN counting as -1."""
    rng = np.random.default_rng(6)
    H, W = 9, 11
    img = rng.random((H, W, 1), dtype=np.float32)
    sp = np.array([np.nan, np.inf, -np.inf, 1e30, -1e30, 3e9, -3e9, 6.8e7, 67108860.0, 65536.0, -65536.0, W - 1, W - 0.5,
                   -0.5, -1.0, -1.5, W + 0.25, 1 / 64, 3 / 64], np.float32)
    my = np.array([2.25], np.float32)
    with np.errstate(all="ignore"):
        out = O.remap_bilinear(img, sp, my, mode)[0, :, 0]
    assert np.isfinite(out).all()
    row = O.remap_bilinear(img, np.arange(W, dtype=np.float32), my, mode)[0, :, 0]       # the blended row itself
    first, last = row[0], row[W - 1]
    if mode == "cv2":
        # 32 * 6.8e7 >= 2^31 -> INT_MIN -> pixel 0; 32 * 67108860 = 2147483520 fits -> far right -> last pixel;
        # (-0.5, -1.5, W + 0.25: both taps on the edge pixel but fractional table weights, p*w0 + p*w1 -- within an ulp of p)
        expect_first = [0, 1, 2, 3, 4, 5, 6, 7, 10, 14, 17]
        expect_last = [8, 9, 11, 12]
        np.testing.assert_allclose(out[[13, 15]], first, rtol=2e-7)
        np.testing.assert_allclose(out[16], last, rtol=2e-7)
    else:
        expect_first = [0, 2, 4, 6, 10, 13, 14, 15]
        expect_last = [1, 3, 5, 7, 8, 9, 11, 12, 16]
    assert np.array_equal(out[expect_first], np.full(len(expect_first), first))
    assert np.array_equal(out[expect_last], np.full(len(expect_last), last))
    # the plain-C oracle follows the same conventions
    assert np.array_equal(C.remap_bilinear(img, sp, my, "hwc", mode)[0, :, 0], out)
    # uint8 sources too, and along the other axis
    img8 = (img * 255).astype(np.uint8)
    with np.errstate(all="ignore"):
        o8 = O.remap_bilinear(np.ascontiguousarray(img8.transpose(1, 0, 2)), my, sp, mode)[:, 0, 0]
    r8 = O.remap_bilinear(np.ascontiguousarray(img8.transpose(1, 0, 2)), my, np.arange(W, dtype=np.float32), mode)[:, 0, 0]
    assert np.array_equal(o8[expect_first], np.full(len(expect_first), r8[0]))
    assert np.array_equal(o8[expect_last], np.full(len(expect_last), r8[W - 1]))
    assert np.array_equal(o8[[13, 15]], [r8[0]] * 2) and o8[16] == r8[W - 1]         # integer weights: exact


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
