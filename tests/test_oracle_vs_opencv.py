"""Pin-on-arrival for the one boundary of the path that cannot be pinned in the build container: OpenCV.

The reference calls ``cv2.remap(img, map_x, map_y, interpolation=cv2.INTER_LINEAR, borderMode=cv2.BORDER_REPLICATE)``
(AGW/new_method.py:268-271, MN/checkpoint_utils.py:195-198) and ``cv2.resize(..., interpolation=cv2.INTER_LINEAR)``
(new_method.py:369).  OpenCV is neither vendored by the reference nor installed here (no network), so
``oracle.warp_oracle.remap_bilinear(mode="cv2")`` / ``resize_linear_cv2`` restate OpenCV's PUBLISHED algorithms
(modules/imgproc/src/imgwarp.cpp: INTER_BITS = 5, ``cvRound(32 m)``, float table weights summed left to right; uint8:
int16 weights x 2^15, ``(sum + 2^14) >> 15``; resize.cpp: 11-bit coefficients) and the HIP kernels are bit-exact against
that restatement -- one reading of the algorithm, three implementations.

This file runs the moment anyone has ``cv2``: it skips without it (``pytest.importorskip``), and with it compares the
restatement with the real functions bit for bit, printing which OpenCV build that was.  `python
tests/golden/make_golden.py --with-opencv` additionally writes ``tests/golden/remap_cv2.npz`` from the same calls, so that the
pin can travel to machines without OpenCV afterwards.  CPU only."""
import numpy as np
import pytest

from oracle import warp_oracle as O

cv2 = pytest.importorskip("cv2", reason="OpenCV is not installed: cv2.remap / cv2.resize stay pinned to the published algorithm only")


def _build_line():
    info = cv2.getBuildInformation()
    keep = [l.strip() for l in info.splitlines() if any(k in l for k in ("Version control", "CPU/HW features", "Baseline:", "Dispatched code"))]
    return f"OpenCV {cv2.__version__}; " + " | ".join(keep)


@pytest.fixture(scope="module", autouse=True)
def announce():
    print("\n[test_oracle_vs_opencv] comparing the restatement against", _build_line())


def cv_remap(img, mx, my):
    X, Y = np.meshgrid(mx.astype(np.float32), my.astype(np.float32))          # what the reference builds (indexing="xy")
    return cv2.remap(img, X, Y, interpolation=cv2.INTER_LINEAR, borderMode=cv2.BORDER_REPLICATE)


def cdf_maps(rng, H, W, Ho, Wo):
    def sm(n, peak):
        lg = rng.standard_normal(n).astype(np.float32); lg[peak[0]:peak[1]] += 3.0
        e = np.exp(lg - lg.max()); return (e / e.sum()).astype(np.float32)[None]
    Fx = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(sm(24, (3, 6)), W), 0))
    Fy = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(sm(24, (10, 13)), H), 0))
    mx, my = O.maps_from_cdf(Fx, Fy, (Ho, Wo))
    return mx[0], my[0]


CASES = [(48, 64, 52, 72), (336, 336, 500, 500), (100, 683, 90, 500), (33, 47, 40, 31), (512, 512, 512, 512)]


@pytest.mark.parametrize("shape", CASES)
@pytest.mark.parametrize("C", [1, 2, 3, 4])
@pytest.mark.parametrize("dt", [np.uint8, np.float32, np.float64])
def test_remap_cdf_maps_bit_exact_vs_cv2(shape, C, dt):
    H, W, Ho, Wo = shape
    rng = np.random.default_rng(H * 31 + W + C)
    img = rng.integers(0, 256, (H, W, C), dtype=np.uint8) if dt == np.uint8 else rng.random((H, W, C)).astype(dt)
    if C == 1:
        img = img[:, :, 0]
    mx, my = cdf_maps(rng, H, W, Ho, Wo)
    got = O.remap_bilinear(img, mx, my, "cv2")
    want = cv_remap(img, mx, my)
    assert got.dtype == want.dtype and np.array_equal(got.reshape(want.shape), want)


@pytest.mark.parametrize("dt", [np.uint8, np.float32, np.float64])
def test_remap_wild_and_non_finite_maps_bit_exact_vs_cv2(dt):
    """Unsorted / out-of-range coordinates, the cvRound ties at 1/64 pixel, NaN / +-Inf / huge values (x86 builds: the
    product becomes INT_MIN -> pixel 0 with a zero fraction; other builds may differ -- the assertion message says which
    build this is)."""
    rng = np.random.default_rng(7)
    H, W, C = 48, 64, 3
    img = rng.integers(0, 256, (H, W, C), dtype=np.uint8) if dt == np.uint8 else rng.random((H, W, C)).astype(dt)
    mx = (rng.random(72) * (W + 6) - 3).astype(np.float32)
    my = (rng.random(52) * (H + 6) - 3).astype(np.float32)
    mx[:8] = np.array([0.015625, 0.046875, 3.0, 62.984375, 63.0, 63.5, -7.25, 1e9], np.float32)
    my[:6] = np.array([0.015625, 47.0, 46.984375, 47.515625, -1e9, 5.5], np.float32)
    assert np.array_equal(O.remap_bilinear(img, mx, my, "cv2"), cv_remap(img, mx, my)), _build_line()
    mx[8:12] = np.array([np.nan, np.inf, -np.inf, 3e38], np.float32)
    my[6:9] = np.array([np.nan, -np.inf, 7e37], np.float32)
    assert np.array_equal(O.remap_bilinear(img, mx, my, "cv2"), cv_remap(img, mx, my)), _build_line()


def test_remap_uint8_saturated_corner_weights_vs_cv2():
    """(ky, kx) = (0, 0): OpenCV saturates the int16 weight 32768 to 32767; (p * 32767 + 2^14) >> 15 == p for every byte,
    so the saturation is not observable -- checked on integer coordinates over all 256 values."""
    img = np.arange(256, dtype=np.uint8).reshape(16, 16)
    mx = np.arange(16, dtype=np.float32); my = np.arange(16, dtype=np.float32)
    assert np.array_equal(O.remap_bilinear(img, mx, my, "cv2"), cv_remap(img, mx, my))
    assert np.array_equal(cv_remap(img, mx, my), img)


@pytest.mark.parametrize("dt", [np.uint8, np.float32])
@pytest.mark.parametrize("io", [((37, 53), (106, 74)), ((37, 53), (30, 20)), ((37, 53), (100, 11)), ((37, 53), (7, 90)),
                                ((36, 52), (26, 18)), ((336, 336), (500, 500)), ((768, 1024), (24, 24)), ((37, 53), (53, 37))])
def test_resize_linear_bit_exact_vs_cv2(dt, io):
    (H, W), (wo, ho) = io
    rng = np.random.default_rng(H + W + wo)
    for C in (1, 3, 4):
        img = rng.integers(0, 256, (H, W, C), dtype=np.uint8) if dt == np.uint8 else rng.random((H, W, C)).astype(dt)
        want = cv2.resize(img, (wo, ho), interpolation=cv2.INTER_LINEAR)
        got = O.resize_linear_cv2(img, (wo, ho))
        assert np.array_equal(got.reshape(want.shape), want), (C, _build_line())
