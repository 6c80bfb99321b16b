"""Property tests of the CPU oracle (hypothesis): the invariants the reference's docstrings promise and the
ones the HIP kernels rely on (monotone maps, CDF shape, right-inverse consistency).  CPU only."""
import numpy as np
import pytest
from hypothesis import example, given, settings, strategies as st

from oracle import warp_oracle as O

# derandomize: the same examples on every run and every clone (no dependence on a local .hypothesis database)
SET = dict(max_examples=40, deadline=None, derandomize=True, database=None)


@settings(**SET)
@given(L=st.integers(24, 700), seed=st.integers(0, 2**31 - 1), peak=st.floats(0.1, 6.0))
def test_right_inverse_pools_back_to_input(L, seed, peak):
    """adaptive_avg_pool1d(x_hat, 24) == y (MN/checkpoint_utils.py:70-72)."""
    rng = np.random.default_rng(seed)
    z = rng.standard_normal((2, 24)) * peak
    y = (np.exp(z) / np.exp(z).sum(1, keepdims=True)).astype(np.float32)
    x = O.upsample_pdf_right_inverse(y, L)
    A = O.pooling_matrix(24, L, np.float64)
    np.testing.assert_allclose(x.astype(np.float64) @ A.T, y, rtol=0, atol=5e-6)
    # mass is preserved up to rounding: sum(x) = sum over windows... (each x is a window average spread)
    assert x.shape == (2, L) and np.isfinite(x).all()


@settings(**SET)
@given(L=st.integers(2, 600), seed=st.integers(0, 2**31 - 1), zero_frac=st.floats(0.0, 0.9))
@example(L=157, seed=117271, zero_frac=0.8984375)     # the cumsum reaches 1.0000001 in front of the forced last element
def test_cdf_shape_and_map_monotone(L, seed, zero_frac):
    """What ``cdf_from_density`` guarantees (MN/checkpoint_utils.py:36-40): a running sum of non-negative terms, so
    non-decreasing -- EXCEPT at the last element, which is overwritten with 1.0 after a float32 cumsum that may have
    reached 1 + 2^-23: a -1 ulp step at the tail.  That dip is what sends ``warp_from_cdf_torch`` into its tie-break ramp
    (:181-184); the GPU parity case built from the pinned example is ``test_cdf_tail_dip_takes_the_tie_break_ramp``."""
    rng = np.random.default_rng(seed)
    p = rng.random((3, L)).astype(np.float32)
    p[rng.random((3, L)) < zero_frac] = 0.0
    F = O.cdf_from_density(p)
    assert F.dtype == np.float32 and (F[:, -1] == 1.0).all()
    assert (np.diff(F[:, :-1], axis=1) >= 0).all() and (F >= 0).all()
    assert (F[:, :-1] <= np.float32(1.0) + np.float32(2.0 ** -23)).all()
    assert (np.diff(F, axis=1)[:, -1] >= -np.float32(2.0 ** -23)).all()
    n_out = int(rng.integers(1, 700))
    m = np.stack([O.axis_map_from_cdf(F[b], n_out) for b in range(3)])
    assert (np.diff(m, axis=1) >= 0).all()                      # non-decreasing: what remap_rows_kernel exploits
    assert (m >= 0).all() and (m <= L).all()
    assert np.all(m[:, 0] >= 0)


def tail_dip_cdf():
    """The CDFs of the pinned example above: row 2 holds 1.0000001 in front of the forced 1.0."""
    L, seed, zero_frac = 157, 117271, 0.8984375
    rng = np.random.default_rng(seed)
    p = rng.random((3, L)).astype(np.float32)
    p[rng.random((3, L)) < zero_frac] = 0.0
    return p, O.cdf_from_density(p)


def test_tail_dip_example_is_a_tail_dip():
    p, F = tail_dip_cdf()
    assert F[2, -2] > 1.0 and F[2, -1] == 1.0          # the input of the GPU case really has the dip
    for n_out in (135, 157, 500):
        xn = np.concatenate([[0.0], F[2].astype(np.float64)]) * n_out
        xn[-1] = n_out
        assert (np.diff(xn) <= 0).any()                # -> the ramp branch of checkpoint_utils.py:181-184


@settings(**SET)
@given(h=st.integers(1, 40), w=st.integers(1, 40), nw=st.integers(1, 80), nh=st.integers(1, 80),
       seed=st.integers(0, 2**31 - 1), tr=st.sampled_from(["identity", "square", "sqrt"]))
def test_attention_maps_monotone_and_in_range(h, w, nw, nh, seed, tr):
    rng = np.random.default_rng(seed)
    att = rng.integers(0, 256, (h, w), dtype=np.uint8)
    mx, my = O.maps_from_attention(att, nw, nh, tr)
    assert mx.shape == (nw,) and my.shape == (nh,)
    assert (np.diff(mx) >= 0).all() and (np.diff(my) >= 0).all()
    assert mx[0] == 0 and my[0] == 0 and mx.max() <= w and my.max() <= h


@settings(**SET)
@given(h=st.integers(2, 24), w=st.integers(2, 24), c=st.integers(1, 4), seed=st.integers(0, 2**31 - 1))
def test_remap_convexity_and_constant(h, w, c, seed):
    """Bilinear resampling is a convex combination: outputs stay within the input range; a constant image
    stays constant for ANY maps (also out-of-range and unsorted ones)."""
    rng = np.random.default_rng(seed)
    img = rng.random((h, w, c), dtype=np.float32)
    mx = (rng.random(17) * (w + 4) - 2).astype(np.float32)
    my = (rng.random(13) * (h + 4) - 2).astype(np.float32)
    for mode in ("exact", "cv2"):
        out = O.remap_bilinear(img, mx, my, mode)
        assert out.shape == (13, 17, c)
        assert out.min() >= img.min() - 1e-6 and out.max() <= img.max() + 1e-6
        const = np.full_like(img, 0.3125)
        assert np.array_equal(O.remap_bilinear(const, mx, my, mode), np.full((13, 17, c), 0.3125, np.float32))
    u8 = (img * 255).astype(np.uint8)
    o8 = O.remap_bilinear(u8, mx, my)
    assert o8.dtype == np.uint8 and o8.min() >= u8.min() and o8.max() <= u8.max()


@settings(**SET)
@given(n=st.integers(1, 2000), seed=st.integers(0, 2**31 - 1))
def test_lanczos_identity_and_range(n, seed):
    """Pillow resample: same size is a copy; up-sampling a constant image keeps the constant."""
    rng = np.random.default_rng(seed)
    v = int(rng.integers(0, 256))
    img = np.full((24, 24), v, np.uint8)
    size = int(rng.integers(24, 200))
    assert (O.lanczos_resize_u8(img, size, size) == v).all()
    r = rng.integers(0, 256, (24, 24), dtype=np.uint8)
    assert np.array_equal(O.lanczos_resize_u8(r, 24, 24), r)


def test_attn_reduce_rows_sum_to_one():
    rng = np.random.default_rng(0)
    lg = rng.standard_normal((3, 2, 8, 640)).astype(np.float32)
    e = np.exp(lg - lg.max(-1, keepdims=True)); rows = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    m = O.attn_reduce_stack(rows, [30, 50])
    np.testing.assert_allclose(m.sum(1), 1.0, atol=2e-6)       # every head renormalised over the image tokens


def test_resize_linear_cv2_properties():
    """The restatement of cv2.resize(INTER_LINEAR) (AGW/new_method.py:369; parity unpinned, OpenCV is absent): same size
    = copy, an exact 2 x 2 decimation = the rounded 2 x 2 mean (INTER_AREA), otherwise within one grey level (uint8: the
    11-bit coefficients and the >> 4 / >> 16 truncations) of float64 bilinear interpolation at half-pixel centres."""
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(O.resize_linear_cv2(img, (53, 37)), img)
    even = img[:36, :52]
    s = even.astype(np.int64)
    assert np.array_equal(O.resize_linear_cv2(even, (26, 18)), ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2))
    for (wo, ho) in [(106, 74), (30, 20), (100, 11), (7, 90), (54, 37)]:
        fx = np.clip((np.arange(wo) + 0.5) * 53 / wo - 0.5, 0, 52); fy = np.clip((np.arange(ho) + 0.5) * 37 / ho - 0.5, 0, 36)
        x0 = np.floor(fx).astype(int); x1 = np.minimum(x0 + 1, 52); y0 = np.floor(fy).astype(int); y1 = np.minimum(y0 + 1, 36)
        tx = (fx - x0)[None, :, None]; ty = (fy - y0)[:, None, None]; f = img.astype(np.float64)
        ref = (f[y0][:, x0] * (1 - tx) + f[y0][:, x1] * tx) * (1 - ty) + (f[y1][:, x0] * (1 - tx) + f[y1][:, x1] * tx) * ty
        assert np.abs(O.resize_linear_cv2(img, (wo, ho)).astype(np.float64) - ref).max() < 1.0
        g = O.resize_linear_cv2((img / 255).astype(np.float32), (wo, ho))
        assert g.dtype == np.float32 and np.abs(g - ref / 255).max() < 5e-6      # float32 coordinates and weights


def test_remap_float64_keeps_double_precision():
    """float64 images are resampled in double (warp_from_cdf_torch's dtype pass-through, MN/checkpoint_utils.py:152,203)."""
    rng = np.random.default_rng(4)
    d = rng.random((9, 11, 2))
    mx = np.sort(rng.random(14).astype(np.float32) * 12 - 1); my = np.sort(rng.random(10).astype(np.float32) * 10 - 1)
    for mode in ("exact", "cv2"):
        a = O.remap_bilinear(d, mx, my, mode)
        b = O.remap_bilinear(d.astype(np.float32), mx, my, mode)
        assert a.dtype == np.float64 and b.dtype == np.float32 and 0 < np.abs(a - b).max() < 1e-6
    assert np.array_equal(O.remap_bilinear(d, np.arange(11, dtype=np.float32), np.arange(9, dtype=np.float32), "cv2"), d)


# ---- the three-step division of the 16-bit attention reduce (attn_f32v.hpp, SharedDiv::first_refinement) ----------------
def _three_step_division_mismatches(dtype: str, d_stride: int, d_offset: int = 0) -> tuple:
    """Emulates q1 = fma(a - d*q0, r, q0), q0 = a*r, for every numerator a and every `d_stride`-th denominator d of the
    shared-division box, with the refined reciprocal r off by -1 / 0 / +1 float32 ulp from RN(1/d) (whatever v_rcp_f32
    returned), and compares RN_T(q1) with RN_T(RN_32(a / d)) -- what `a / d` in the model dtype gives in the reference.
    float64 / 80-bit long double hold every intermediate exactly, so each fma rounds once.  -> (pairs checked, mismatches)."""
    ld = np.longdouble
    assert np.finfo(ld).nmant >= 63, "needs the x87 80-bit long double"
    if dtype == "float16":
        a16 = np.arange(0, 0x7C00, dtype=np.uint16).view(np.float16)             # every non-negative finite float16
        d16 = np.arange(1, 0x4401, dtype=np.uint16).view(np.float16)             # (0, 4]
        a = a16.astype(np.float32)
        dens = d16.astype(np.float32)
        def rn_t(x32):
            with np.errstate(over="ignore"):                                      # quotients above 65504 become inf, as on the GPU
                return x32.astype(np.float16).view(np.uint16)
    else:
        hi = np.arange(0x0D80, 0x4980 + 1, dtype=np.uint32)                      # bfloat16 patterns of 2^-100 .. 2^20
        a = np.concatenate([np.zeros(1, np.float32), (hi << 16).view(np.float32)])
        dens = (np.arange(0x2180, 0x4080 + 1, dtype=np.uint32) << 16).view(np.float32)   # 2^-60 .. 4
        def rn_t(x32):
            b = x32.view(np.uint32).astype(np.uint64)
            return ((b + 0x7FFF + ((b >> 16) & 1)) >> 16).astype(np.uint32)
    a64 = a.astype(np.float64)
    checked = bad = 0
    for d in dens[d_offset::d_stride]:
        ref = rn_t(a / d)                                                         # numpy float32 division is IEEE
        r_rn = np.float32(1.0) / d
        d64 = np.float64(d)
        for r in (np.nextafter(r_rn, np.float32(0)), r_rn, np.nextafter(r_rn, np.float32(np.inf))):
            r64 = np.float64(r)
            q0 = (a64 * r64).astype(np.float32)                                   # 11 x 24 bits: exact in float64
            e1 = (a64 - d64 * q0.astype(np.float64)).astype(np.float32)           # exact (cancellation), representable
            q1 = (q0.astype(ld) + e1.astype(ld) * ld(r)).astype(np.float32)       # <= 61 bits: exact in long double
            bad += int(np.count_nonzero(rn_t(q1) != ref))
            checked += a.size
    return checked, bad


def test_three_step_division_float16():
    """A stride sample of the exhaustive run (every 24th denominator; `python tests/test_oracle_properties.py` runs all)."""
    checked, bad = _three_step_division_mismatches("float16", 24, 7)
    assert checked > 60_000_000 and bad == 0


def test_three_step_division_bfloat16():
    checked, bad = _three_step_division_mismatches("bfloat16", 12, 5)
    assert checked > 20_000_000 and bad == 0


if __name__ == "__main__":
    for dt in ("float16", "bfloat16"):
        print(dt, "exhaustive (pairs x 3 reciprocals, mismatches):", _three_step_division_mismatches(dt, 1), flush=True)
