"""GPU parity tests: every HIP stage, called through the C ABI (ctypes) by the Python surface that
mirrors the reference, against the CPU oracle on the same seeded inputs, against the golden vectors
captured from the reference, and -- at BASELINE's full sizes -- through size-independent properties.

Bars: bit-exact wherever the oracle defines the operation order (everything except the places
listed in DESIGN.md "parity"); <= 1e-4 max-abs per pixel end to end (north_star tolerance)."""
import os

import numpy as np
import pytest
import torch

from attwarp_amd import _lib
from oracle import warp_oracle as O
from conftest import pool_input, clip_input, clip_digest, config1_inputs, main_batched_loop_inputs

pytestmark = pytest.mark.gpu

TOL_PIXEL = 1e-4   # north_star: max-abs per-pixel tolerance, float32 images in [0,1]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from attwarp_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "HIP library missing -- the product has no fallback"
    _lib.load()
    return torch.device("cuda:0")


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def for_each(items, fn, workers=None):
    """fn(item) for every item on a pool of host threads (numpy and the ctypes C oracle release the GIL): the full-size
    tests check EVERY image of a batch against the oracle, not a sample."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=workers or min(16, os.cpu_count() or 1)) as ex:
        list(ex.map(fn, items))


def softmax_rows(rng, shape, peak=None):
    lg = rng.standard_normal(shape).astype(np.float32)
    if peak is not None:
        lg[..., peak[0]:peak[1]] += 3.0
    e = np.exp(lg - lg.max(-1, keepdims=True))
    return (e / e.sum(-1, keepdims=True)).astype(np.float32)


# =============================== A1 / A2 =====================================
def test_attn_steps_vs_oracle_and_golden(dev, golden):
    from attwarp_amd import attention_extraction as ae
    g = golden("attn_reduce")
    starts, ends = [int(v) for v in g["starts"]], [int(v) for v in g["ends"]]
    hl = ae.BatchMaskHookLogger(model=None, device=dev)
    hl.set_batch_image_token_ranges(starts, ends)
    for t in range(4):
        full = g[f"step_in_{t}"]                                   # [B, heads, 1, kv]
        hl._process_attention(T(full, dev))
        got = N(hl.step_attentions[-1])
        assert np.array_equal(got, O.attn_reduce_step(full, starts, ends))          # bit-exact vs oracle
        # float32 tree accumulation here, torch's CPU float32 order in the fixture: a few ulps (one of the 6912
        # fixture values at 3.15e-7, see tests/test_oracle_golden.py)
        np.testing.assert_allclose(got, g[f"step_out_{t}"], rtol=4e-7, atol=0)
    maps = hl.finalize_batch()
    assert len(maps) == 3 and all(m.shape == (24, 24) for m in maps)
    fin = np.stack([N(m) for m in maps])
    assert np.array_equal(fin.reshape(3, 576), O.attn_finalize([N(s) for s in hl.step_attentions]))
    np.testing.assert_allclose(fin, g["final"], rtol=3e-7, atol=0)
    hl.reinit()
    assert hl.step_attentions == [] and hl.batch_size == 0


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_attn_step_16bit_rows_every_alignment(dev, dt):
    """The 16-bit rows are read as aligned four-token words and put back together in registers: every residue of the slice
    start (prompt lengths 32 .. 39 and 0 .. 3), of the row stride (kv lengths that are and are not multiples of four) and
    of the tensor's own offset inside its allocation (a view 0 .. 3 elements into a buffer), against the oracle bit for
    bit; the last row ends at the very end of the buffer."""
    from attwarp_amd import attention_extraction as ae
    rng = np.random.default_rng(77)
    heads, ntok = 32, 576
    for kv in (640, 613, 579):
        for off in (0, 1, 2, 3):
            B = 9
            a = rng.random((B, heads, 2, kv), dtype=np.float32) ** 6
            a /= a.sum(-1, keepdims=True)
            starts = np.array([(s0 % (kv - ntok + 1)) for s0 in (32, 33, 34, 35, 36, 37, 0, 1, 3)], np.int32)
            buf = torch.zeros(off + a.size, device=dev, dtype=dt)
            view = buf[off:].view(B, heads, 2, kv)
            view.copy_(T(a, dev).to(dt))
            got = ae.attn_reduce_step(view, T(starts, dev), ntok)
            if dt == torch.float16:
                ref = O.attn_reduce_step(N(view), starts, starts + ntok)
                assert np.array_equal(N(got), ref), (kv, off)
            else:       # numpy has no bfloat16: the generic one-token-per-lane kernel (strided call) is the cross-check
                wide = torch.zeros(B, heads, 2, 2 * kv, device=dev, dtype=dt)
                wide[..., ::2] = view
                ref = ae.attn_reduce_step(wide[..., ::2], T(starts, dev), ntok)
                assert torch.equal(got, ref), (kv, off)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_attn_step_16bit_rows_full_three_vectors(dev, dt):
    """ntok == 768 fills all three 256-token vectors of the four-tokens-per-lane kernel: lane 63 of the last vector owns
    tokens 764..767, and with an odd slice start its fourth token would have to come from a chunk no vector loads (ADVICE r3:
    the realigned load returned token 511 there).  Odd and even starts, odd and even row strides, with and without room
    behind the slice, against the oracle (float16) / the one-token-per-lane kernel (bfloat16)."""
    from attwarp_amd import attention_extraction as ae
    rng = np.random.default_rng(78)
    heads, ntok = 8, 768
    for kv in (800, 801, 771, 768):
        B = 6
        a = rng.random((B, heads, 1, kv), dtype=np.float32) ** 4
        a /= a.sum(-1, keepdims=True)
        starts = np.array([s0 % (kv - ntok + 1) for s0 in (0, 1, 2, 3, 31, 32)], np.int32)
        for off in (0, 1):
            buf = torch.zeros(off + a.size, device=dev, dtype=dt)
            view = buf[off:].view(B, heads, 1, kv)
            view.copy_(T(a, dev).to(dt))
            got = ae.attn_reduce_step(view, T(starts, dev), ntok)
            if dt == torch.float16:
                assert np.array_equal(N(got), O.attn_reduce_step(N(view), starts, starts + ntok)), (kv, off)
            else:
                wide = torch.zeros(B, heads, 1, 2 * kv, device=dev, dtype=dt)
                wide[..., ::2] = view
                assert torch.equal(got, ae.attn_reduce_step(wide[..., ::2], T(starts, dev), ntok)), (kv, off)


@pytest.mark.parametrize("dt", [torch.float32, torch.float16, torch.bfloat16])
def test_attn_step_tail_lengths_and_head_counts(dev, dt):
    """512 < ntok <= 576 (the third 256-token vector of a row carries <= 64 tokens): every tail length, head counts that do
    and do not fill the last group of heads in flight (1 .. 33), odd and even slice starts and row strides, slices with and
    without room behind them, rows with zeros / values far below the shared division's range -- against the oracle bit for
    bit (float32 / float16) or the one-token-per-lane kernel (bfloat16).  (Written for the round-4 shared-tail-vector
    experiment, docs/experiments.md; kept because no other test varies the head count.)"""
    from attwarp_amd import attention_extraction as ae
    rng = np.random.default_rng(79)
    for ntok in (516, 540, 572, 576):
        for heads in (1, 3, 5, 13, 16, 32, 33):
            kv = ntok + (7 if heads % 2 else 4)
            B = 4
            a = rng.random((B, heads, 1, kv), dtype=np.float32) ** 5
            a[0, :, :, ::7] = 0.0                                      # zeros among the numerators
            a[1, heads // 2] *= 1e-30                                  # one head far below the shared division's range
            a /= np.maximum(a.sum(-1, keepdims=True), 1e-30)
            starts = np.array([s0 % (kv - ntok + 1) for s0 in (0, 1, 3, 7)], np.int32)
            for off in (0, 1):
                buf = torch.zeros(off + a.size, device=dev, dtype=dt)
                view = buf[off:].view(B, heads, 1, kv)
                view.copy_(T(a, dev).to(dt))
                got = ae.attn_reduce_step(view, T(starts, dev), ntok)
                if dt == torch.bfloat16:
                    wide = torch.zeros(B, heads, 1, 2 * kv, device=dev, dtype=dt)
                    wide[..., ::2] = view
                    assert torch.equal(got, ae.attn_reduce_step(wide[..., ::2], T(starts, dev), ntok)), (ntok, heads, off)
                else:
                    ref = O.attn_reduce_step(N(view), starts, starts + ntok)
                    assert np.array_equal(N(got).view(np.uint8), ref.view(np.uint8)), (ntok, heads, off)


def test_axis_kernels_long_axis_dynamic_lds(dev):
    """An axis of 8192 pixels needs ~100 KB of LDS per workgroup: above the 64 KB a launch gets by default, so the launchers
    grant it to the kernel first.  Same results as the oracle chain; the stream entry points either grant it
    (attn_reduce_and_maps) or refuse up front (warp_step_fused: small images only)."""
    from attwarp_amd import pipeline
    rng = np.random.default_rng(5)
    B, L = 2, 8192
    px = rng.random((B, 24), dtype=np.float32); px /= px.sum(-1, keepdims=True)
    py = rng.random((B, 24), dtype=np.float32); py /= py.sum(-1, keepdims=True)
    mx, my = pipeline.axis_maps_from_pdf(T(px, dev), T(py, dev), (L, L // 2))
    Fx = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(px, L // 2), 0))
    Fy = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(py, L), 0))
    rx, ry = O.maps_from_cdf(Fx, Fy)
    assert np.array_equal(N(mx), rx) and np.array_equal(N(my), ry)
    steps = torch.rand(3, B, 576, device=dev)
    sx, sy = pipeline.axis_maps_from_attention_steps(steps, (L, 64))
    att = O.attn_finalize([N(s) for s in steps]).reshape(B, 1, 24, 24)
    qx, qy = O.gt_marginals(att)
    Gx = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(qx, 64), 0))
    Gy = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(qy, L), 0))
    gx, gy = O.maps_from_cdf(Gx, Gy)
    assert np.array_equal(N(sx), gx) and np.array_equal(N(sy), gy)


def test_attn_step_division_matches_ieee_everywhere(dev):
    """The float32 step kernel divides by (row sum + 1e-12) with a reciprocal shared by the row when the operands sit
    inside the box where v_div_scale does not rescale (attn.hip, SharedDiv) and with the plain IEEE division outside:
    rows of real softmax values, rows with exponents on and beyond the box edges (2^-100, 2^-103, denormals, 2^20,
    1e30), zeros, negative values, tiny and huge row sums, Inf and NaN -- numpy's float32 division bit for bit."""
    from attwarp_amd import attention_extraction as ae
    rng = np.random.default_rng(2024)
    B, heads, kv, ntok = 6, 32, 640, 576
    a = rng.random((B, heads, 1, kv), dtype=np.float32)
    a[0] = np.exp(rng.normal(0, 6, (heads, 1, kv))).astype(np.float32)
    a[0] /= a[0].sum(-1, keepdims=True)                                     # a real softmax: wide dynamic range
    edge = np.array([2.0 ** -100, np.nextafter(np.float32(2.0 ** -100), np.float32(0)), 2.0 ** -103, 1e-40, 1e-45, 0.0,
                     2.0 ** 20, np.nextafter(np.float32(2.0 ** 20), np.float32(np.inf)), 1e30, 3e-39], np.float32)
    a[1, :, 0, 40:40 + ntok] = rng.choice(edge, (heads, ntok))                # sums dominated by 2^20 / 1e30 entries
    a[1, ::2, 0, 40:40 + ntok] = np.minimum(a[1, ::2, 0, 40:40 + ntok], np.float32(1.0))
    a[2] = (a[2] * np.float32(1e-22)).astype(np.float32)                      # row sums ~ 3e-20 < 2^-60: IEEE path
    a[2, 5] = a[2, 5] * np.float32(1e-12)                                     # ... and below 1e-12: den = 1e-12
    a[3, :, 0, ::7] = 0.0
    a[3, 3, 0, 100] = -0.25                                                   # negative numerator
    a[3, 4, 0, 101] = np.inf
    a[3, 6, 0, 102] = np.nan
    a[4] = rng.choice(np.array([0.0, 1e-30, 1.0, 0.5, 3.0], np.float32), (heads, 1, kv))
    a[5, :, 0, :] = np.float32(2.0 ** -100) * rng.integers(1, 1 << 20, (heads, kv)).astype(np.float32)
    starts = [40, 40, 35, 64, 0, 17]
    ends = [st + ntok for st in starts]
    hl = ae.BatchMaskHookLogger(model=None, device=dev)
    hl.set_batch_image_token_ranges(starts, ends)
    hl._process_attention(T(a, dev))
    got = N(hl.step_attentions[-1])
    with np.errstate(all="ignore"):
        ref = O.attn_reduce_step(a, starts, ends)
    assert np.array_equal(got, ref, equal_nan=True)
    assert np.isfinite(ref[[0, 2, 4, 5]]).all() and not np.isfinite(ref[3]).all()
    # the generic kernel (any dtype / stride; plain IEEE division) on the same kind of input: float16 rows with subnormal
    # halves, 65504, zeros, a negative, Inf, NaN and rows whose sum rounds to 0 or overflows in float16 ...
    h = rng.random((B, heads, 1, kv), dtype=np.float32).astype(np.float16)
    h[0] = (np.exp(rng.normal(0, 4, (heads, 1, kv))) * 1e-3).astype(np.float16)
    h[1, :, 0, :] = rng.choice(np.array([0.0, 6e-8, 6.1e-5, 1.0, 65504.0, 0.333], np.float16), (heads, kv))
    h[2] = np.float16(6e-8) * (rng.random((heads, 1, kv)) < 0.01)                    # sums of a few subnormals
    h[2, 7] = 0                                                                     # all-zero row: 0 / 0
    h[3, 3, 0, 100] = np.float16(-0.25)
    h[3, 4, 0, 101] = np.float16(np.inf)
    h[3, 6, 0, 102] = np.float16(np.nan)
    h[4] = np.float16(60000.0) * (rng.random((heads, 1, kv)) < 0.5)                  # row sums overflow float16
    hl.reinit()
    hl.set_batch_image_token_ranges(starts, ends)
    hl._process_attention(T(h, dev))
    with np.errstate(all="ignore"):
        ref16 = O.attn_reduce_step(h, starts, ends)
    assert np.array_equal(N(hl.step_attentions[-1]), ref16, equal_nan=True)
    # ... and float32 rows with a non-unit kv stride (a transposed view): generic kernel, float32 division
    at = np.ascontiguousarray(a.transpose(0, 1, 3, 2))                               # [B, heads, kv, 1]
    hl.reinit()
    hl.set_batch_image_token_ranges(starts, ends)
    hl._process_attention(T(at, dev).transpose(2, 3))
    assert np.array_equal(N(hl.step_attentions[-1]), ref, equal_nan=True)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_attn_steps_low_precision(dev, golden, dt):
    from attwarp_amd import attention_extraction as ae
    g = golden("attn_reduce")
    starts, ends = [int(v) for v in g["starts"]], [int(v) for v in g["ends"]]
    hl = ae.BatchMaskHookLogger(model=None, device=dev)
    hl.set_batch_image_token_ranges(starts, ends)
    for t in range(4):
        a = T(g[f"step_in_{t}"], dev).to(dt)
        hl._process_attention(a)
        assert hl.step_attentions[-1].dtype == dt
        if dt == torch.float16:
            ref = O.attn_reduce_step(N(a), starts, ends)
            assert np.array_equal(N(hl.step_attentions[-1]), ref)                      # bit-exact in fp16
            np.testing.assert_allclose(ref.astype(np.float32), g[f"step_out16_{t}"].astype(np.float32), rtol=2e-3)
    fin = torch.stack(hl.finalize_batch())
    assert fin.dtype == dt
    if dt == torch.float16:
        np.testing.assert_allclose(N(fin).astype(np.float32), g["final16"].astype(np.float32), rtol=2e-3)
    else:
        np.testing.assert_allclose(N(fin.float()), g["final"], rtol=2e-2)


# ===================== "next" row 4: probe from (query, key) ==================
@pytest.mark.parametrize("name", ["prefill", "decode"])
def test_probe_vs_oracle_and_golden(dev, golden, name):
    from attwarp_amd import attention_extraction as ae
    g = golden("attn_probe")
    starts, pads, ntok, sc = g[f"{name}_starts"], g[f"{name}_pads"], int(g[f"{name}_ntok"]), float(g[f"{name}_scaling"])
    for tag, dt in (("f32", np.float32), ("f16", np.float16)):
        q, k = g[f"{name}_q_last"].astype(dt), g[f"{name}_key"].astype(dt)
        got = N(ae.probe_last_query(T(q, dev), T(k, dev), T(starts.astype(np.int32), dev), ntok,
                                    T(pads.astype(np.int32), dev), sc))
        assert got.dtype == dt
        assert np.array_equal(got, O.attn_probe_step(q, k, starts, ntok, sc, pads))          # bit-exact vs oracle
        # vs transformers' eager attention + the reference's hook: fp16 exact, fp32 within torch's sgemm/expf
        if dt == np.float16:
            assert np.array_equal(got, g[f"{name}_{tag}_step"])
        else:
            np.testing.assert_allclose(got, g[f"{name}_{tag}_step"], rtol=2e-6)


@pytest.mark.parametrize("dt,D,H,Hkv,kv", [
    (np.float16, 128, 32, 32, 640),      # LLaVA-1.5-7B geometry
    (np.float16, 128, 8, 2, 701),        # grouped-query attention, ragged kv
    (np.float32, 128, 4, 4, 333),        # two 16-byte chunks per lane
    (np.float16, 64, 6, 3, 97),          # 8 lanes per row
    (np.float16, 96, 4, 4, 130),         # 4 lanes per row, 3 chunks per lane
    (np.float32, 20, 3, 1, 64),          # 1 lane per row, 5 chunks -> unsupported
    (np.float16, 256, 2, 1, 1500),       # 2 chunks per lane
])
def test_probe_shapes_and_strides(dev, dt, D, H, Hkv, kv):
    from attwarp_amd import attention_extraction as ae, _lib
    rng = np.random.default_rng(D * 7 + kv)
    B, ntok = 3, min(576, kv - 40)
    q = rng.standard_normal((B, H, D)).astype(dt)
    k = rng.standard_normal((B, Hkv, kv, D)).astype(dt)
    starts = np.array([5, 17, 40], np.int32)
    pads = np.array([0, 3, 39], np.int32)
    sc = D ** -0.5
    if D == 20:
        with pytest.raises(_lib.AttWarpError, match="chunks per lane"):
            ae.probe_last_query(T(q, dev), T(k, dev), T(starts, dev), ntok, T(pads, dev))
        return
    ref = O.attn_probe_step(q, k, starts, ntok, sc, pads)
    got = N(ae.probe_last_query(T(q, dev), T(k, dev), T(starts, dev), ntok, T(pads, dev)))
    assert np.array_equal(got, ref)
    # the layout HF hands over at prefill without a cache: [B,kv,Hkv,D] memory viewed as [B,Hkv,kv,D],
    # query [B,q,H,D] memory viewed as [B,H,q,D]; the last row is taken by the wrapper
    k_t = T(np.ascontiguousarray(k.transpose(0, 2, 1, 3)), dev).transpose(1, 2)
    q_full = rng.standard_normal((B, 3, H, D)).astype(dt)
    q_full[:, -1] = q
    q_t = T(q_full, dev).transpose(1, 2)
    assert Hkv == 1 or not k_t.is_contiguous()
    got2 = N(ae.probe_last_query(q_t, k_t, T(starts, dev), ntok, T(pads, dev)))
    assert np.array_equal(got2, ref)
    # no padding argument == all-zero padding
    ref0 = O.attn_probe_step(q, k, starts, ntok, sc, None)
    assert np.array_equal(N(ae.probe_last_query(T(q, dev), T(k, dev), T(starts, dev), ntok)), ref0)


def test_probe_bf16_and_misaligned_inputs(dev):
    from attwarp_amd import attention_extraction as ae
    rng = np.random.default_rng(5)
    B, H, kv, D, ntok = 2, 8, 300, 128, 192
    q = torch.from_numpy(rng.standard_normal((B, H, D)).astype(np.float32)).to(dev).bfloat16()
    k = torch.from_numpy(rng.standard_normal((B, H, kv, D)).astype(np.float32)).to(dev).bfloat16()
    starts = torch.tensor([7, 30], dtype=torch.int32, device=dev)
    got = ae.probe_last_query(q, k, starts, ntok)
    assert got.dtype == torch.bfloat16
    # bf16 has no numpy dtype: compare with the float32 oracle on the same (bf16-representable) inputs
    ref = O.attn_probe_step(N(q.float()), N(k.float()), N(starts), ntok, D ** -0.5, None)
    np.testing.assert_allclose(N(got.float()), ref, rtol=3e-2)
    # a key view whose rows start 2 bytes off a 16-byte boundary is copied by the wrapper, not mis-read
    big = torch.zeros(B, H, kv, D + 8, device=dev, dtype=torch.float16)
    kh = k.half()
    big[..., 1:D + 1] = kh
    view = big[..., 1:D + 1]
    a = ae.probe_last_query(q.half(), view, starts, ntok)
    b = ae.probe_last_query(q.half(), kh, starts, ntok)
    assert torch.equal(a, b)


@pytest.mark.parametrize("layer,rtol", [(0, 4e-3), (1, 5e-2)])
def test_probe_equals_hooked_eager_on_hf_llama(dev, layer, rtol):
    """End to end through the installed transformers' LlamaAttention: the reference's way (eager attention,
    output_attentions forced on the target layer, hook on the probabilities) against register_probe on the
    model's fast path.  Layer 0 sees identical inputs on both paths (tolerance = fp16 rounding of the
    GEMM-vs-exact logits); deeper layers add the eager-vs-sdpa drift of the hidden states."""
    transformers = pytest.importorskip("transformers")
    from attwarp_amd import attention_extraction as ae
    cfg = transformers.LlamaConfig(hidden_size=512, intermediate_size=1024, num_hidden_layers=2,
                                   num_attention_heads=4, num_key_value_heads=4, vocab_size=1000, head_dim=128,
                                   max_position_embeddings=2048)
    torch.manual_seed(3)
    model = transformers.LlamaForCausalLM(cfg).to(dev).half().eval()
    B, L, ntok = 3, 640, 576
    pads = [0, 5, 11]
    ids = torch.randint(0, 1000, (B, L), device=dev)
    am = torch.ones(B, L, dtype=torch.long, device=dev)
    for b, p_ in enumerate(pads):
        am[b, :p_] = 0
    nxt = torch.randint(0, 1000, (3, B, 1), device=dev)               # fixed continuation: both paths see the same tokens
    starts = [30 + p_ for p_ in pads]

    def run(impl, probe):
        model.config._attn_implementation = impl
        hl = ae.BatchMaskHookLogger(model, dev, layer_index=layer)
        hl.set_batch_image_token_ranges(starts, [s + ntok for s in starts])
        hl.register_probe() if probe else hl.register_hook_and_patch()
        try:
            with torch.no_grad():
                out = model(input_ids=ids, attention_mask=am, use_cache=True)
                cache, mask = out.past_key_values, am
                for t in range(3):
                    mask = torch.cat([mask, torch.ones(B, 1, dtype=torch.long, device=dev)], dim=1)
                    out = model(input_ids=nxt[t], attention_mask=mask, past_key_values=cache, use_cache=True)
                    cache = out.past_key_values
        finally:
            hl.remove_hook_and_unpatch()
        return hl

    hooked = run("eager", probe=False)
    probed = run("sdpa", probe=True)
    assert len(hooked.step_attentions) == len(probed.step_attentions) == 4
    for a, b in zip(hooked.step_attentions, probed.step_attentions):
        assert a.shape == b.shape == (B, ntok) and a.dtype == b.dtype == torch.float16
        np.testing.assert_allclose(N(b.float()), N(a.float()), rtol=rtol, atol=2e-6)
    fa, fb = torch.stack(hooked.finalize_batch()), torch.stack(probed.finalize_batch())
    np.testing.assert_allclose(N(fb.float()), N(fa.float()), rtol=rtol, atol=2e-6)


def test_attn_strided_prefill_and_empty(dev):
    from attwarp_amd import attention_extraction as ae
    rng = np.random.default_rng(21)
    B, heads, q, kv = 3, 8, 5, 700
    big = softmax_rows(rng, (B, heads, q, kv + 9))
    a = T(big, dev)[..., 4:4 + kv]                         # non-contiguous view (kv stride 1, row stride kv+9)
    starts = [3, 40, 100]
    hl = ae.BatchMaskHookLogger(None, dev)
    assert [tuple(m.shape) for m in (hl.finalize_batch())] == []           # batch_size 0 -> empty list
    hl.set_batch_image_token_ranges(starts, [s + 576 for s in starts])
    empty = hl.finalize_batch()
    assert len(empty) == 3 and empty[0].shape == (576,)                    # flat uniform map, reference :404-408
    assert torch.allclose(empty[0], torch.full((576,), 1 / 576, device=dev))
    hl._process_attention(a)
    ref = O.attn_reduce_step(big[..., 4:4 + kv], starts, [s + 576 for s in starts])
    assert np.array_equal(N(hl.step_attentions[0]), ref)
    # clipped range: end beyond kv shortens every slice equally
    hl2 = ae.BatchMaskHookLogger(None, dev)
    hl2.set_batch_image_token_ranges([150, 150, 150], [726, 726, 726])
    hl2._process_attention(a)
    assert hl2.step_attentions[0].shape == (3, 550)
    assert np.array_equal(N(hl2.step_attentions[0]), O.attn_reduce_step(big[..., 4:4 + kv], [150] * 3, [726] * 3))
    # ragged clipped ranges cannot be stacked (reference raises in torch.stack)
    hl3 = ae.BatchMaskHookLogger(None, dev)
    hl3.set_batch_image_token_ranges([100, 150, 150], [676, 726, 726])
    with pytest.raises(RuntimeError, match="equal size"):
        hl3._process_attention(a)


def test_attn_out_of_range_starts_are_clamped(dev):
    """A slice start outside [0, kv - ntok] must not read outside the attention row: it is clamped on the
    device; ntok > kv is rejected on the host."""
    from attwarp_amd import attention_extraction as ae
    rng = np.random.default_rng(12)
    B, heads, kv = 4, 8, 600
    rows = softmax_rows(rng, (2, B, heads, kv))
    bad = np.array([-5, 0, 24, 10 ** 6], np.int32)
    ok = np.clip(bad, 0, kv - 576).astype(np.int32)
    a = N(ae.attn_reduce_stack(T(rows, dev), T(bad, dev)))
    b = N(ae.attn_reduce_stack(T(rows, dev), T(ok, dev)))
    assert np.array_equal(a, b)
    assert np.array_equal(b, O.attn_reduce_stack(rows, ok))
    s1 = N(ae.attn_reduce_step(T(rows[0][:, :, None, :], dev), T(bad, dev), 576))
    assert np.array_equal(s1, N(ae.attn_reduce_step(T(rows[0][:, :, None, :], dev), T(ok, dev), 576)))
    with pytest.raises(_lib.AttWarpError, match="ntok"):
        ae.attn_reduce_stack(T(rows[..., :500], dev), T(ok, dev))
    # the hook loggers hold the ranges as Python lists: a range outside the attention row is the caller's bookkeeping
    # bug and raises on the host instead of being clamped into attention for other tokens
    hl = ae.BatchMaskHookLogger(model=None, device=dev)
    hl.set_batch_image_token_ranges([0, -5], [576, 571])
    with pytest.raises(ValueError, match="outside the attention row"):
        hl._process_attention(T(rows[0][:2, :, None, :], dev))
    ml = ae.MaskHookLogger(None, dev)
    ml.image_token_start, ml.image_token_end = -3, 573
    with pytest.raises(ValueError, match="outside the attention row"):
        ml._process_attention(T(rows[0][:1, :, None, :], dev))


def test_attn_stack_fused_and_single_logger(dev):
    from attwarp_amd import attention_extraction as ae
    rng = np.random.default_rng(22)
    T_, B, heads, kv = 5, 4, 32, 640
    rows = softmax_rows(rng, (T_, B, heads, kv), peak=(200, 260))
    starts = np.array([35, 36, 42, 37], np.int32)
    got = N(ae.attn_reduce_stack(T(rows, dev), T(starts, dev)))
    assert np.array_equal(got, O.attn_reduce_stack(rows, starts))
    # MaskHookLogger: default range [1, 577), mean over steps AND batch rows
    ml = ae.MaskHookLogger(None, dev)
    for t in range(T_):
        ml._process_attention(T(rows[t][:, :, None, :], dev))
    steps = [O.attn_reduce_step(rows[t][:, :, None, :], [1] * B, [577] * B) for t in range(T_)]
    cat = np.concatenate(steps, 0)                                   # [T*B, 576] (reference: torch.cat)
    ref = O.attn_finalize([r[None] for r in cat])[0]                 # mean over all rows
    assert np.array_equal(N(ml.finalize()), ref)


# =============================== A3 / A4 =====================================
def test_revise_mask(dev, golden):
    from attwarp_amd import attention_extraction as ae
    g = golden("mask_postproc")
    got = N(ae.revise_mask(T(g["masks"], dev), 3, 10))
    ref = O.revise_mask(g["masks"], 3, 10)
    assert np.abs(got - ref).max() <= 6e-8                               # <= 1 ulp near 1.0 vs oracle
    np.testing.assert_allclose(got, g["revised"], rtol=0, atol=5e-7)      # vs reference (torch float32 reductions)
    one = N(ae.revise_mask(T(g["masks"][0], dev), kernel_size=5, enhance_coe=4))
    assert one.shape == (24, 24)
    np.testing.assert_allclose(one, g["mask0_k5_c4"], rtol=0, atol=5e-7)
    with pytest.raises(AssertionError):
        ae.revise_mask(T(g["masks"][0], dev), kernel_size=4)
    # a constant map: normalize("min") divides 0 by 0 and torch carries the NaN through sigmoid, clamp and the box
    # filter (the up-sampled uint8 mask is then all 0 and the warp the identity) -- found by tests/fuzz/fuzz_stages.py
    flat = np.full((2, 24, 24), 0.3, np.float32)
    flat[1] = g["masks"][0]
    with np.errstate(all="ignore"):
        ref_flat = O.revise_mask(flat, 3, 10)
    got_flat = N(ae.revise_mask(T(flat, dev), 3, 10))
    assert np.isnan(ref_flat[0]).all() and np.isnan(got_flat[0]).all()
    assert np.abs(got_flat[1] - ref_flat[1]).max() <= 6e-8
    up = N(ae.upsample_mask_lanczos(T(got_flat, dev), (64, 48)))
    assert not up[0].any() and up[1].any()


@pytest.mark.parametrize("variant", [-1, 1, 2])    # column-strip kernel (up-sampling) / two-kernel form / row-block fused kernel
@pytest.mark.parametrize("wh", [(336, 336), (500, 375), (1024, 1024), (24, 48), (17, 24), (24, 24)])
def test_mask_upsample_lanczos_bit_exact(dev, golden, wh, variant):
    from attwarp_amd import attention_extraction as ae
    g = golden("mask_postproc")
    w, h = wh
    with _lib.debug_override(lanczos_variant=variant):
        got_f = N(ae.upsample_mask_lanczos(T(g["revised"][:2], dev), (w, h)))      # float mask: x255 truncation inside
        got_u = N(ae.upsample_mask_lanczos(T(g["u8"][:2], dev), (w, h)))
    if wh == (24, 24):
        assert np.array_equal(got_f, g["u8"][:2]) and np.array_equal(got_u, g["u8"][:2])
    else:
        ref = g[f"lanczos_{w}x{h}"]                                            # real Pillow output
        assert got_f.shape == ref.shape
        assert np.array_equal(got_f, ref) and np.array_equal(got_u, ref)


def test_blend_mask_surface(dev, golden):
    from PIL import Image
    from attwarp_amd import attention_extraction as ae
    g = golden("mask_postproc")
    img = Image.fromarray(np.zeros((375, 500, 3), np.uint8))
    merged, mask = ae.blend_mask(img, T(g["masks"][1], dev), 10, 3, Image.LANCZOS, 0)
    assert mask.mode == "L" and mask.size == (500, 375)
    rev_u8 = O.mask_to_u8(N(ae.revise_mask(T(g["masks"][1], dev), 3, 10)))
    assert np.array_equal(np.array(mask), O.lanczos_resize_u8(rev_u8, 500, 375))
    with pytest.raises(NotImplementedError):
        ae.blend_mask(3.14, T(g["masks"][1], dev), 10, 3, Image.LANCZOS, 0)


def test_adaptive_avg_pool2d_drop_in(dev, golden):
    """F.adaptive_avg_pool2d's call shape (MN/trainer.py:197,433,465) + the trainer's sanitise (:202)."""
    from attwarp_amd import pipeline
    g = golden("pool_marginals")
    A = pool_input(336)                                                   # [2,1,336,336]
    out = pipeline.adaptive_avg_pool2d(T(A, dev), (24, 24))
    assert out.shape == (2, 1, 24, 24)
    assert np.array_equal(N(out), O.adaptive_avg_pool24(A))
    np.testing.assert_allclose(N(out), g["P_336"], rtol=3e-6)
    # 3-D input, non-square output, torch's own result on the GPU as a cross-check
    rng = np.random.default_rng(3)
    X = rng.standard_normal((3, 45, 70)).astype(np.float32)
    got = pipeline.adaptive_avg_pool2d(T(X, dev), (7, 12))
    ref = torch.nn.functional.adaptive_avg_pool2d(T(X, dev)[:, None], (7, 12))[:, 0]
    assert got.shape == (3, 7, 12)
    np.testing.assert_allclose(N(got), N(ref), rtol=2e-6, atol=1e-7)
    # sanitise: NaN / inf windows and negative means become 0, everything else is untouched
    Y = X.copy()
    Y[0, :10, :10] = np.nan
    Y[1, 20:, 30:] = np.inf
    Y[2] = -np.abs(Y[2])
    raw = N(pipeline.adaptive_avg_pool2d(T(Y, dev), (7, 12)))
    san = N(pipeline.adaptive_avg_pool2d(T(Y, dev), (7, 12), sanitize=True))
    expect = np.maximum(np.nan_to_num(raw, nan=0.0, posinf=0.0, neginf=0.0), 0)
    assert np.array_equal(san, expect) and np.isnan(raw).any() and np.isinf(raw).any() and not san[2].any()


# =============================== A5 / A6 / A7 ================================
@pytest.mark.parametrize("S", [336, 512, 1024])
def test_adaptive_pool_and_marginals(dev, golden, S):
    from attwarp_amd import checkpoint_utils as cu, _lib
    from attwarp_amd._lib import call, ptr, stream_ptr
    g = golden("pool_marginals")
    A = pool_input(S)
    a = T(A[:, 0], dev)
    out = torch.empty(2, 24, 24, device=dev)
    call("attwarp_adaptive_avg_pool", ptr(a), 2, S, S, 24, 24, 0, ptr(out), stream_ptr(dev))
    assert np.array_equal(N(out), O.adaptive_avg_pool24(A)[:, 0])
    np.testing.assert_allclose(N(out), g[f"P_{S}"][:, 0], rtol=3e-6)
    px, py = cu.gt_marginals(T(g[f"P_{S}"], dev))
    pxo, pyo = O.gt_marginals(g[f"P_{S}"])
    assert np.array_equal(N(px), pxo) and np.array_equal(N(py), pyo)
    np.testing.assert_allclose(N(px), g[f"px_{S}"], rtol=3e-7, atol=1e-9)
    # full-resolution marginals (trainer.py:357,501,578)
    pxf, pyf = cu.gt_marginals(T(A, dev))
    pxfo, pyfo = O.gt_marginals(A)
    assert np.array_equal(N(pxf), pxfo) and np.array_equal(N(pyf), pyfo)


def test_gt_marginals_negative_nonsquare(dev, golden):
    from attwarp_amd import checkpoint_utils as cu
    g = golden("pool_marginals")
    px, py = cu.gt_marginals(T(g["Afull"], dev))
    pxo, pyo = O.gt_marginals(g["Afull"])
    assert np.array_equal(N(px), pxo) and np.array_equal(N(py), pyo)
    np.testing.assert_allclose(N(px), g["pxf"], rtol=5e-7, atol=1e-9)
    np.testing.assert_allclose(N(py), g["pyf"], rtol=5e-7, atol=1e-9)


def test_safe_softmax(dev, golden):
    from attwarp_amd import model
    g = golden("pool_marginals")
    got = N(model.safe_softmax(T(g["logits"], dev), dim=1, eps=1e-6))
    assert np.abs(got - O.safe_softmax(g["logits"])).max() <= 1.2e-7
    np.testing.assert_allclose(got, g["safe_softmax"], rtol=5e-7, atol=1e-12)
    assert np.isfinite(got).all()
    # dim handling: softmax over dim 0 of the transposed input
    got0 = N(model.safe_softmax(T(g["logits"].T.copy(), dev), dim=0))
    assert np.array_equal(got0.T, got)


def test_marginalnet_forward_gpu(dev, golden):
    from attwarp_amd import model
    g = golden("marginalnet")
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd|")}
    net = model.MarginalNet(32, 48, hidden=16).eval()
    net.load_state_dict(sd)
    net = net.to(dev)
    with torch.no_grad():
        px, py = net(T(g["fmap"], dev), 24, 24, T(g["ttok"], dev), T(g["tmask"], dev))
    np.testing.assert_allclose(N(px), g["px"], rtol=1e-4, atol=1e-6)      # convs run on MIOpen/rocBLAS
    np.testing.assert_allclose(N(py), g["py"], rtol=1e-4, atol=1e-6)
    assert np.allclose(N(px).sum(1), 1, atol=1e-5)


def test_marginalnet_hidden256_config5_and_real_checkpoint(dev, golden, tmp_path):
    """BASELINE configs[4] at its stated size: MarginalNet(1024, 4096, hidden=256) on seeded weights / inputs against
    the reference's own forward (tests/golden/marginalnet_full.npz, generated by importing MN/model.py), loaded through
    a REAL ``torch.save({"epoch", "model", "opt", "cfg"})`` file in the trainer's format (MN/trainer.py:660-683),
    then through the device-resident chain ``warp_from_marginalnet`` (maps bit-exact vs the oracle given px, py)."""
    from conftest import marginalnet_full_state, marginalnet_full_inputs
    from attwarp_amd import model, pipeline
    g = golden("marginalnet_full")
    net = model.MarginalNet(1024, 4096, hidden=256).eval()
    sd = marginalnet_full_state({k: tuple(v.shape) for k, v in net.state_dict().items()})
    assert sum(v.numel() for v in sd.values()) == int(g["n_params"]) == 2755074
    assert abs(sum(v.double().sum().item() for v in sd.values()) - float(g["sd_checksum"])) < 1e-6
    opt = torch.optim.AdamW(model.MarginalNet(1024, 4096, hidden=256).parameters(), lr=3e-4, weight_decay=1e-4)
    path = str(tmp_path / "marginal_net_epoch_1.pt")
    torch.save({"epoch": 1, "model": sd, "opt": opt.state_dict(),
                "cfg": {"seed": 13, "epochs": 50, "batch_size": 128, "lr": 3e-4, "wd": 1e-4, "grad_clip": 1.0, "workers": 4,
                        "image_size": 512, "num_per_ds": 12000, "hidden": 256, "w_cdf": 10.0, "axis_len": 256,
                        "llava_model": "liuhaotian/llava-v1.5-7b"}}, path)
    model.load_reference_checkpoint(net, path)
    net = net.to(dev)
    fmap, ttok, tmask = (t.to(dev) for t in marginalnet_full_inputs(4))
    cap = {}
    hs = [net.head_x.register_forward_pre_hook(lambda m, a: cap.__setitem__("vx", a[0].detach())),
          net.head_y.register_forward_pre_hook(lambda m, a: cap.__setitem__("vy", a[0].detach())),
          net.txt_pool.register_forward_pre_hook(lambda m, a: cap.__setitem__("tmean", a[0].detach()))]
    img = torch.rand(4, 3, 336, 336, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    out, px, py = pipeline.warp_from_marginalnet(net, fmap, ttok, tmask, img)
    for h in hs:
        h.remove()
    # masked token mean has no GEMM in front of it: tight; everything behind MIOpen / rocBLAS: GEMM-order tolerance
    np.testing.assert_allclose(N(cap["tmean"][:, ::64]), g["tmean_sub"], rtol=0, atol=3e-7)
    np.testing.assert_allclose(N(cap["vx"][:, ::16, ::3]), g["vx_sub"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(N(cap["vy"][:, ::16, ::3]), g["vy_sub"], rtol=2e-4, atol=2e-5)
    assert abs(float(cap["vx"].double().sum()) - float(g["sum_vx"])) < 2e-3
    np.testing.assert_allclose(N(px), g["px"], rtol=5e-4, atol=1e-7)
    np.testing.assert_allclose(N(py), g["py"], rtol=5e-4, atol=1e-7)
    assert np.allclose(N(px).sum(1), 1, atol=1e-5) and np.allclose(N(py).sum(1), 1, atol=1e-5)
    # the rest of the chain is exact given (px, py)
    Fx = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(N(px), 336), 0))
    Fy = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(N(py), 336), 0))
    assert np.array_equal(N(out), O.warp_from_cdf(N(img), Fx, Fy))
    # and end to end against the reference's (px, py): the warp moves by less than north_star's tolerance per pixel
    Fxr = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(g["px"], 336), 0))
    Fyr = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(g["py"], 336), 0))
    mxr, myr = O.maps_from_cdf(Fxr, Fyr)
    mxg, myg = O.maps_from_cdf(Fx, Fy)
    assert np.abs(mxr - mxg).max() < 0.05 and np.abs(myr - myg).max() < 0.05       # source coordinates, pixels


def test_training_path_is_differentiable(dev, golden):
    """the drop-ins sit on the autograd path of the reference's training loss (MN/trainer.py:210-260:
    ``px_s, py_s = net(...)``, ``upsample_pdf_right_inverse(px_s, W)``, loss.backward()).  With grad needed the
    same maths runs on differentiable ops (or an autograd.Function around the kernel): gradients reach every
    parameter, values equal the inference (HIP) path, and the right-inverse gradient equals the explicit matrix."""
    from attwarp_amd import model, checkpoint_utils as cu
    g = golden("marginalnet")
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd|")}
    net = model.MarginalNet(32, 48, hidden=16)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    args = (T(g["fmap"], dev), 24, 24, T(g["ttok"], dev), T(g["tmask"], dev))
    px, py = net(*args)                                        # grad mode on, parameters require grad
    assert px.requires_grad and py.requires_grad
    with torch.no_grad():
        px0, py0 = net(*args)                                  # HIP tail + HIP safe_softmax
    np.testing.assert_allclose(N(px), N(px0), rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(N(py), N(py0), rtol=2e-5, atol=1e-7)
    W = 336
    px_img = cu.upsample_pdf_right_inverse(px, W).clamp_min(0)
    Fx = cu.cdf_from_density(px_img)
    target = torch.linspace(0, 1, W, device=dev)[None].expand_as(Fx)
    loss = ((Fx - target) ** 2).mean() + (cu.upsample_pdf_right_inverse(py, W) ** 2).mean()
    loss.backward()
    missing = [n for n, p in net.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
    assert not missing, missing
    assert any(float(p.grad.abs().max()) > 0 for p in net.parameters())
    # right-inverse: forward == kernel, backward == explicit M^T with M = inv @ A
    y = torch.rand(3, 24, device=dev, requires_grad=True)
    out = cu.upsample_pdf_right_inverse(y, 500)
    with torch.no_grad():
        assert torch.equal(out, cu.upsample_pdf_right_inverse(y.detach(), 500))
    w = torch.randn(3, 500, device=dev)
    (out * w).sum().backward()
    A = torch.zeros(24, 500, dtype=torch.float64)
    for k in range(24):
        s0, e0 = (k * 500) // 24, -((-(k + 1) * 500) // 24)
        A[k, s0:e0] = 1.0 / (e0 - s0)
    M = torch.from_numpy(O.right_inverse_core(24, 500)) @ A          # [24, 500]
    np.testing.assert_allclose(N(y.grad), (w.double().cpu() @ M.T).numpy(), rtol=1e-5, atol=1e-7)
    # the other CDF helpers: differentiable route equals the kernels' values
    p = torch.rand(2, 64, device=dev, requires_grad=True)
    np.testing.assert_allclose(N(cu.cdf_from_density(p)), N(cu.cdf_from_density(p.detach())), rtol=0, atol=2e-7)
    Fd = cu.resample_cdf(cu.cdf_from_density(p), 100)
    np.testing.assert_allclose(N(Fd), N(cu.resample_cdf(cu.cdf_from_density(p.detach()), 100)), rtol=0, atol=5e-7)
    Fd.sum().backward()
    assert p.grad is not None and torch.isfinite(p.grad).all()
    Aq = torch.rand(2, 1, 12, 9, device=dev, requires_grad=True)
    mx, my = cu.gt_marginals(Aq)
    mx0, my0 = cu.gt_marginals(Aq.detach())
    np.testing.assert_allclose(N(mx), N(mx0), rtol=1e-6, atol=1e-8)
    lg = torch.randn(4, 24, device=dev, requires_grad=True)
    np.testing.assert_allclose(N(model.safe_softmax(lg)), N(model.safe_softmax(lg.detach())), rtol=2e-6, atol=1e-8)


# =============================== A8 / A9 / A10 / A11 ==========================
@pytest.mark.parametrize("L", [336, 500, 512, 1024])
def test_pdf_cdf_chain(dev, golden, L):
    from attwarp_amd import checkpoint_utils as cu
    g = golden("pdf_cdf")
    x = N(cu.upsample_pdf_right_inverse(T(g["y"], dev), L))
    assert np.array_equal(x, O.upsample_pdf_right_inverse(g["y"], L))
    np.testing.assert_allclose(x, g[f"x_{L}"], rtol=0, atol=4e-7 * np.abs(g[f"x_{L}"]).max())
    A = O.pooling_matrix(24, L, np.float64)
    np.testing.assert_allclose(x.astype(np.float64) @ A.T, g["y"], atol=2e-6)          # docstring invariant
    p = np.maximum(g[f"x_{L}"], 0)
    F = N(cu.cdf_from_density(T(p, dev)))
    assert np.array_equal(F, O.cdf_from_density(p))
    np.testing.assert_allclose(F, g[f"cdf_{L}"], rtol=6e-7, atol=0)
    assert (np.diff(F, axis=1) >= 0).all() and (F[:, -1] == 1).all()


def test_right_inverse_shapes_and_bad_density(dev, golden):
    from attwarp_amd import checkpoint_utils as cu
    g = golden("pdf_cdf")
    assert np.array_equal(N(cu.upsample_pdf_right_inverse(T(g["y"][0], dev), 336)),
                          O.upsample_pdf_right_inverse(g["y"][0], 336))
    x3 = cu.upsample_pdf_right_inverse(T(g["y"].reshape(1, 3, 24), dev), 336)
    assert x3.shape == (1, 3, 336)
    assert np.array_equal(N(x3), O.upsample_pdf_right_inverse(g["y"].reshape(1, 3, 24), 336))
    F = N(cu.cdf_from_density(T(g["p_bad"], dev)))
    assert np.array_equal(F, O.cdf_from_density(g["p_bad"]))
    assert np.abs(F - g["cdf_bad"]).max() <= 2.4e-7


def test_cdf_scan_paths_agree_with_sequential_oracle(dev):
    """cdf_from_density scans in parallel when every partial sum is provably exact in double and falls back
    to one sequential lane otherwise (tiny densities); both must equal the oracle's sequential cumsum."""
    from attwarp_amd import checkpoint_utils as cu
    rng = np.random.default_rng(71)
    for L in (37, 336, 1024, 4099):
        p = rng.random((4, L)).astype(np.float32)
        p[1] *= rng.random(L).astype(np.float32) ** 8          # wide dynamic range, still >= 2^-28 after normalising?
        p[2, ::3] = 1e-13                                       # forces the sequential path
        p[3] = 0; p[3, L // 2] = 1.0; p[3, 5] = 3e-39           # one spike + a denormal
        F = N(cu.cdf_from_density(T(p, dev)))
        assert np.array_equal(F, O.cdf_from_density(p)), L


def test_strictly_increasing_and_resample(dev, golden):
    from attwarp_amd import checkpoint_utils as cu
    g = golden("pdf_cdf")
    m = N(cu._make_strictly_increasing(T(g["F24"], dev)))
    assert np.array_equal(m, O.make_strictly_increasing(g["F24"]))
    assert np.abs(m - g["msi"]).max() <= 2.4e-7
    for L in (336, 1024):
        r = N(cu.resample_cdf(T(g["F24"], dev), L))
        assert np.array_equal(r, O.resample_cdf(g["F24"], L))
        np.testing.assert_allclose(r, g[f"resample_{L}"], rtol=0, atol=3e-7)
        assert (np.diff(r, axis=1) > 0).all()


@pytest.mark.parametrize("case", ["sq336", "rect", "to500", "sq1024", "ties"])
def test_maps_from_cdf_bit_exact_vs_reference(dev, golden, case):
    from attwarp_amd import checkpoint_utils as cu
    g = golden("maps_from_cdf")
    Fx = g["ties_F"] if case == "ties" else g[f"{case}_Fx"]
    Fy = g["ties_F"] if case == "ties" else g[f"{case}_Fy"]
    out = tuple(int(v) for v in g[f"{case}_out"])
    mx, my = cu.axis_maps_from_cdf(T(Fx, dev), T(Fy, dev), out)
    assert np.array_equal(N(mx), g[f"{case}_mx"])            # the float32 maps the REFERENCE hands to cv2.remap
    assert np.array_equal(N(my), g[f"{case}_my"])


def test_maps_non_monotone_cdf_replays_numpy_search(dev):
    """A garbage (non-monotone, NaN) 'CDF' makes np.interp's result depend on its search path; the
    kernel replays that path exactly."""
    from attwarp_amd import checkpoint_utils as cu
    rng = np.random.default_rng(31)
    F = rng.random((3, 40)).astype(np.float32)
    F[1, 7] = np.nan
    F[2] = np.sort(F[2])[::-1]
    ref_x, ref_y = O.maps_from_cdf(F, F, (50, 60))
    mx, my = cu.axis_maps_from_cdf(T(F, dev), T(F, dev), (50, 60))
    assert np.array_equal(N(mx), ref_x, equal_nan=True) and np.array_equal(N(my), ref_y, equal_nan=True)


def test_cdf_tail_dip_takes_the_tie_break_ramp(dev):
    """cdf_from_density overwrites the last element with 1.0 after a float32 cumsum that can reach 1 + 2^-23
    (MN/checkpoint_utils.py:39-40): a -1 ulp step at the tail, which sends warp_from_cdf_torch into its tie-break ramp
    (:181-184).  The pinned hypothesis example of tests/test_oracle_properties.py through attwarp_cdf_from_density ->
    attwarp_axis_map_from_cdf, and 24-bin PDFs whose up-sampled density has the same dip through the fused
    attwarp_axis_maps_from_pdf: bit-exact against the oracle."""
    from attwarp_amd import checkpoint_utils as cu, pipeline
    from test_oracle_properties import tail_dip_cdf
    p, F = tail_dip_cdf()
    assert F[2, -2] > 1.0
    Fg = cu.cdf_from_density(T(p, dev))
    assert np.array_equal(N(Fg), F)
    for out in ((135, 135), (157, 157), (500, 336), (64, 700)):
        rx, ry = O.maps_from_cdf(F, F, out)
        mx, my = cu.axis_maps_from_cdf(Fg, Fg, out)
        assert np.array_equal(N(mx), rx) and np.array_equal(N(my), ry), out
    hit = 0
    for seed, L in ((3, 500), (4, 500), (6, 500), (18, 1024), (29, 1024), (58, 1024)):
        rng = np.random.default_rng(seed)
        z = rng.standard_normal((8, 24)) * rng.uniform(0.5, 5)
        px = (np.exp(z) / np.exp(z).sum(1, keepdims=True)).astype(np.float32)
        Fx = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(px, L), 0))
        hit += int((Fx[:, -2] > 1.0).sum())
        py = px[::-1].copy()
        Fy = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(py, L), 0))
        for out in (None, (500, 500)):
            rx, ry = O.maps_from_cdf(Fx, Fy, out)
            mx, my = pipeline.axis_maps_from_pdf(T(px, dev), T(py, dev), (L, L), out)
            assert np.array_equal(N(mx), rx) and np.array_equal(N(my), ry), (seed, L, out)
    assert hit >= 6                                          # the cases really carry the dip


def test_fused_axis_maps_from_pdf(dev):
    from attwarp_amd import pipeline
    rng = np.random.default_rng(32)
    px = softmax_rows(rng, (5, 24)); py = softmax_rows(rng, (5, 24), peak=(3, 6))
    for (H, W, out) in [(336, 336, None), (1024, 1024, None), (48, 100, (60, 80)), (512, 336, (500, 500))]:
        mx, my = pipeline.axis_maps_from_pdf(T(px, dev), T(py, dev), (H, W), out)
        Fx = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(px, W), 0))
        Fy = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(py, H), 0))
        rx, ry = O.maps_from_cdf(Fx, Fy, out)
        assert np.array_equal(N(mx), rx) and np.array_equal(N(my), ry)


# =============================== A13 =========================================
def test_maps_from_attention_vs_reference_golden(dev, golden):
    """All 96 golden combinations (4 attention maps x 6 transforms x inverse x 2 output sizes) against the
    float32 maps the REFERENCE handed to cv2.remap.  Profiles and totals are accumulated in numpy's own
    order (ascending rows for axis=0, pairwise for axis=1 and for the totals), so identity / square / sqrt
    are bit-exact; exp / log go through the device libm (1 ulp in double), where a float32 map entry may
    differ by one ulp."""
    from attwarp_amd import new_method as nm
    g = golden("maps_from_attention")
    n_elem = n_diff = 0
    for key in [str(c) for c in g["combos"]]:
        aname, tr, inv, wh, es, ed = key.split("|")
        nw, nh = (int(v) for v in wh.split("x"))
        att = g[aname]
        a = T(att if att.dtype in (np.uint8, np.float32, np.float64) else att.astype(np.float64), dev)[None]
        mx, my = nm.attention_axis_maps(a, nw, nh, tr, float(es), float(ed), bool(int(inv)))
        for got, ref in ((N(mx)[0], g[f"mx|{key}"]), (N(my)[0], g[f"my|{key}"])):
            if tr in ("identity", "square", "sqrt", "bogus"):
                assert np.array_equal(got, ref, equal_nan=True), key
                continue
            fin = np.isfinite(ref)
            assert np.array_equal(np.isnan(got), np.isnan(ref)), key
            n_elem += fin.sum(); n_diff += (got[fin] != ref[fin]).sum()
            np.testing.assert_allclose(got[fin], ref[fin], rtol=2.5e-7, atol=1e-30, err_msg=key)
    assert n_diff <= 1e-3 * n_elem, (n_diff, n_elem)


def test_kernels_vs_random_reference_cases(dev, golden):
    """The kernels against tests/golden/random_cases.npz -- 100 randomised, partly hostile cases that make_golden.py ran
    through the REFERENCE (no oracle in between): A13 maps bit for bit (exp / log: the device libm, one float32 ulp on
    < 0.1 % of the entries), A9 CDFs within 2.5e-7 and the A11 maps of the reference's own CDFs bit for bit, A3 within
    the float32-vs-float64 reduction tolerance, the constant map NaN for NaN."""
    from attwarp_amd import new_method as nm, checkpoint_utils as cu, attention_extraction as ae
    g = golden("random_cases")
    trs = ["identity", "square", "sqrt", "exp", "log", "bogus"]
    n_elem = n_diff = 0
    for key in [str(k) for k in g["names"]]:
        if key.startswith("a13_"):
            nw, nh, ti, inv, es, ed = g[key + "_par"]
            tr = trs[int(ti)]
            mx, my = nm.attention_axis_maps(T(g[key + "_att"], dev)[None], int(nw), int(nh), tr, float(es), float(ed), bool(inv))
            for got, ref in ((N(mx)[0], g[key + "_mx"]), (N(my)[0], g[key + "_my"])):
                if tr in ("identity", "square", "sqrt", "bogus"):
                    assert np.array_equal(got, ref, equal_nan=True), key
                    continue
                fin = np.isfinite(ref)
                assert np.array_equal(np.isnan(got), np.isnan(ref)), key
                n_elem += fin.sum(); n_diff += (got[fin] != ref[fin]).sum()
                np.testing.assert_allclose(got[fin], ref[fin], rtol=2.5e-7, atol=1e-30, err_msg=key)
        elif key.startswith("a11_"):
            Fx = cu.cdf_from_density(T(g[key + "_p"][None], dev)); Fy = cu.cdf_from_density(T(g[key + "_q"][None], dev))
            np.testing.assert_allclose(N(Fx)[0], g[key + "_Fx"], rtol=0, atol=2.5e-7, err_msg=key)
            np.testing.assert_allclose(N(Fy)[0], g[key + "_Fy"], rtol=0, atol=2.5e-7, err_msg=key)
            mx, my = cu.axis_maps_from_cdf(T(g[key + "_Fx"][None], dev), T(g[key + "_Fy"][None], dev), tuple(int(v) for v in g[key + "_out"]))
            assert np.array_equal(N(mx)[0], g[key + "_mx"], equal_nan=True) and np.array_equal(N(my)[0], g[key + "_my"], equal_nan=True), key
        else:
            ks, coe = g[key + "_par"]
            rev = N(ae.revise_mask(T(g[key + "_m"], dev), int(ks), float(coe)))
            assert np.array_equal(np.isnan(rev), np.isnan(g[key + "_rev"])), key
            np.testing.assert_allclose(rev, g[key + "_rev"], rtol=0, atol=6e-7 * max(1.0, float(coe) / 3), err_msg=key)
    assert n_diff <= 2e-3 * max(n_elem, 1), (n_diff, n_elem)


@pytest.mark.parametrize("hw", [(1024, 1024), (500, 333), (129, 1000), (7, 5)])
def test_maps_from_attention_bit_exact_vs_numpy_order(dev, hw):
    """Sizes whose pairwise tree is irregular (leaf lengths 80/88/..., tails, n < 8)."""
    from attwarp_amd import new_method as nm
    h, w = hw
    rng = np.random.default_rng(h * 7 + w)
    att = rng.integers(0, 256, (2, h, w), dtype=np.uint8)
    attf = (rng.random((2, h, w)) * 3).astype(np.float32)
    for a, tr in ((att, "identity"), (attf, "sqrt"), (attf.astype(np.float64), "square")):
        mx, my = nm.attention_axis_maps(T(a, dev), 500, 400, tr)
        for b in range(2):
            rx, ry = O.maps_from_attention(a[b], 500, 400, tr)
            assert np.array_equal(N(mx)[b], rx) and np.array_equal(N(my)[b], ry), (hw, tr)


@pytest.mark.parametrize("hw", [(1024, 1024), (336, 336), (500, 500), (333, 500), (100, 1000), (64, 260), (70, 132),
                                (40, 2048)])
def test_maps_from_uint8_attention_fast_kernel(dev, hw):
    """uint8 attention (the up-sampled mask of the main_batched chain) through profiles_u8_kernel -- raw bytes in LDS,
    per-thread stride-8 accumulator chains, table look-up for sqrt / exp / log -- against the oracle (numpy's own
    summation orders) for identity / square, and bit-for-bit against the generic kernel for every transform
    (the device libm is the same in both), with and without the inverse, on regular and irregular pairwise trees
    (leaves of 128, 80/88, 120/128/64/68 with tails, one leaf, 16 leaves)."""
    from attwarp_amd import new_method as nm
    h, w = hw
    rng = np.random.default_rng(h * 13 + w)
    att = rng.integers(0, 256, (3, h, w), dtype=np.uint8)
    att[1] = np.clip(rng.normal(128, 3, (h, w)), 0, 255).astype(np.uint8)          # smooth: table look-ups collide
    att[2, : h // 2] = 0                                                           # half empty
    a = T(att, dev)
    for tr in ("identity", "square", "sqrt", "exp", "log"):
        for inv in (False, True):
            kw = dict(transform=tr, exp_scale=1.0, exp_divisor=50.0, apply_inverse=inv)
            mx, my = nm.attention_axis_maps(a, 500, 400, **kw)
            with _lib.debug_override(profiles_variant=1):
                gx, gy = nm.attention_axis_maps(a, 500, 400, **kw)
            # (log + inverse overflows to NaN maps in the reference too: NaNs must match position by position)
            assert np.array_equal(N(mx), N(gx), equal_nan=True) and np.array_equal(N(my), N(gy), equal_nan=True), (hw, tr, inv)
            if tr in ("identity", "square") and not inv:
                for b in range(3):
                    rx, ry = O.maps_from_attention(att[b], 500, 400, tr)
                    assert np.array_equal(N(mx)[b], rx) and np.array_equal(N(my)[b], ry), (hw, tr, b)


@pytest.mark.parametrize("hw", [(1024, 1024), (336, 336), (333, 500), (100, 1000), (70, 132), (40, 2048)])
def test_maps_and_marginals_from_float32_fast_kernel(dev, hw):
    """float32 attention / activation maps through profiles_f32_kernel (raw floats in LDS, the transform in registers,
    row and column sums on different waves): the oracle's maps (numpy's summation orders) for identity / square, the
    generic kernel bit for bit for every transform (variant 2 sends sqrt / exp / log through the new kernel too), and
    gt_marginals against the oracle and the generic kernel -- negative values, NaN and Inf included."""
    from attwarp_amd import new_method as nm, checkpoint_utils as cu
    h, w = hw
    rng = np.random.default_rng(h * 7 + w)
    att = rng.random((3, h, w), dtype=np.float32) * 3.0 - 0.5            # some negatives: clamped
    att[1] = rng.normal(1.0, 1e-3, (h, w)).astype(np.float32)
    att[2, : h // 2] = 0
    a = T(att, dev)
    for tr in ("identity", "square", "sqrt", "exp", "log"):
        for inv in (False, True):
            kw = dict(transform=tr, exp_scale=1.0, exp_divisor=50.0, apply_inverse=inv)
            with _lib.debug_override(profiles_variant=1):
                gx, gy = nm.attention_axis_maps(a, 500, 400, **kw)
            for variant in (-1, 2):
                with _lib.debug_override(profiles_variant=variant):
                    mx, my = nm.attention_axis_maps(a, 500, 400, **kw)
                assert np.array_equal(N(mx), N(gx), equal_nan=True) and np.array_equal(N(my), N(gy), equal_nan=True), (hw, tr, inv, variant)
            if tr in ("identity", "square") and not inv:
                for b in range(3):
                    rx, ry = O.maps_from_attention(att[b], 500, 400, tr)
                    assert np.array_equal(N(mx)[b], rx) and np.array_equal(N(my)[b], ry), (hw, tr, b)
    A = att[:, None].copy()
    A[0, 0, h // 3, w // 5] = np.nan
    A[1, 0, h // 2, w // 2] = np.inf
    px, py = cu.gt_marginals(T(A, dev))
    with _lib.debug_override(profiles_variant=1):
        qx, qy = cu.gt_marginals(T(A, dev))
    with np.errstate(invalid="ignore"):
        pxo, pyo = O.gt_marginals(A)
    assert np.array_equal(N(px), N(qx), equal_nan=True) and np.array_equal(N(py), N(qy), equal_nan=True)
    assert np.array_equal(N(px), pxo, equal_nan=True) and np.array_equal(N(py), pyo, equal_nan=True)


def test_uniform_attention_gives_identity_warp(dev):
    from attwarp_amd import new_method as nm
    rng = np.random.default_rng(33)
    img = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    out = nm.warp_image_by_attention(img, np.full((48, 64), 7, np.uint8), 64, 48, transform="identity")
    assert np.abs(out.astype(int) - img.astype(int)).max() <= 0     # identity map up to 1e-13 -> same pixels


# =============================== A12 remap ===================================
SHAPES = [(336, 336, 336, 336, 3), (64, 96, 50, 70, 3), (33, 47, 40, 31, 1), (128, 128, 128, 128, 4),
          (100, 28, 30, 200, 3), (32, 40, 500, 500, 3), (512, 512, 500, 500, 3), (31, 29, 31, 29, 2)]


def make_maps(rng, B, H, W, Ho, Wo, kind="cdf"):
    if kind == "cdf":
        px = softmax_rows(rng, (B, 24)); py = softmax_rows(rng, (B, 24), peak=(10, 13))
        Fx = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(px, W), 0))
        Fy = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(py, H), 0))
        return O.maps_from_cdf(Fx, Fy, (Ho, Wo))
    if kind == "wild":      # unsorted, out of range, negative: arbitrary caller maps
        mx = (rng.random((B, Wo)) * (W + 6) - 3).astype(np.float32)
        my = (rng.random((B, Ho)) * (H + 6) - 3).astype(np.float32)
        return mx, my
    if kind == "identity":
        return (np.tile(np.arange(Wo, dtype=np.float32) * (W / Wo), (B, 1)),
                np.tile(np.arange(Ho, dtype=np.float32) * (H / Ho), (B, 1)))
    raise ValueError(kind)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("kind", ["cdf", "wild"])
@pytest.mark.parametrize("mode", ["exact", "cv2"])
def test_remap_all_modes_layouts_dtypes(dev, shape, kind, mode):
    """Both arithmetic modes x both kernels (staged rows kernel, generic gather kernel) x both layouts x both dtypes:
    bit-exact against the oracle (cv2: oracle._remap_cv2_compat)."""
    from attwarp_amd import checkpoint_utils as cu
    H, W, Ho, Wo, C = shape
    rng = np.random.default_rng(hash((shape, kind)) % 2**32)
    B = 3
    mx, my = make_maps(rng, B, H, W, Ho, Wo, kind)
    for dt in (np.float32, np.uint8):
        img = rng.random((B, H, W, C), dtype=np.float32)
        if dt == np.uint8:
            img = (img * 255).astype(np.uint8)
        ref = np.stack([O.remap_bilinear(img[b], mx[b], my[b], mode) for b in range(B)])
        # "rows": the staged kernels (uint8 cv2: the integer form); "rows-float": uint8 cv2 on the float-pipeline form;
        # "gather": the generic kernel
        for variant in ("rows", "gather", "rows-float"):
            with _lib.debug_override(remap_variant={"rows": -1, "gather": 1, "rows-float": 2}[variant]):
                hwc = N(cu.remap_separable(T(img, dev), T(mx, dev), T(my, dev), mode=mode, channels_last=True))
                chw = N(cu.remap_separable(T(img.transpose(0, 3, 1, 2), dev), T(mx, dev), T(my, dev), mode=mode))
            assert np.array_equal(hwc, ref), (dt.__name__, variant, "hwc")
            assert np.array_equal(chw.transpose(0, 2, 3, 1), ref), (dt.__name__, variant, "chw")


@pytest.mark.parametrize("shape", [(26, 1024, 18, 1024, 4), (24, 1364, 20, 1000, 3), (25, 340, 30, 1364, 3),
                                   (24, 2048, 11, 2048, 2), (27, 4096, 7, 4096, 1), (30, 700, 9, 256, 3)])
@pytest.mark.parametrize("kind", ["cdf", "wild"])
def test_remap_uint8_cv2_integer_kernel_wide_rows(dev, shape, kind):
    """The integer-arithmetic uint8 cv2 kernel at every (source dwords, output dwords) per-thread count up to its
    4096-byte row limit, interleaved and planar: equal to the oracle, to the float-pipeline form and to the gather
    kernel bit for bit."""
    from attwarp_amd import checkpoint_utils as cu
    H, W, Ho, Wo, C = shape
    rng = np.random.default_rng(H * 7 + W + Wo)
    B = 2
    mx, my = make_maps(rng, B, H, W, Ho, Wo, kind)
    img = rng.integers(0, 256, (B, H, W, C), dtype=np.uint8)
    ref = np.stack([O.remap_bilinear(img[b], mx[b], my[b], "cv2") for b in range(B)])
    for variant in (-1, 2, 1):
        with _lib.debug_override(remap_variant=variant):
            hwc = N(cu.remap_separable(T(img, dev), T(mx, dev), T(my, dev), mode="cv2", channels_last=True))
            chw = N(cu.remap_separable(T(img.transpose(0, 3, 1, 2), dev), T(mx, dev), T(my, dev), mode="cv2"))
        assert np.array_equal(hwc, ref), (variant, "hwc")
        assert np.array_equal(chw.transpose(0, 2, 3, 1), ref), (variant, "chw")


@pytest.mark.parametrize("shape", [(26, 1024, 18, 1024, 3), (25, 336, 7, 500, 3), (24, 64, 3, 64, 4), (30, 700, 2, 256, 1),
                                   (70, 512, 66, 300, 3)])
def test_remap_uint8_integer_kernel_prefetch_depth_and_block_order(dev, shape):
    """The integer uint8 kernel requests its source rows 1, 2 or 4 output rows ahead (register sets) and can order its
    workgroups per XCD, plainly or in groups: every combination, with row blocks of 1 ... 64 rows (fewer rows than
    register sets, remainders of the unroll by four) and several blocks per workgroup, equals the oracle."""
    from attwarp_amd import checkpoint_utils as cu
    H, W, Ho, Wo, C = shape
    rng = np.random.default_rng(H * 13 + W + Wo)
    B = 3
    mx, my = make_maps(rng, B, H, W, Ho, Wo, "cdf")
    img = rng.integers(0, 256, (B, H, W, C), dtype=np.uint8)
    ref = np.stack([O.remap_bilinear(img[b], mx[b], my[b], "cv2") for b in range(B)])
    ti, tx, ty = T(img, dev), T(mx, dev), T(my, dev)
    tc = T(img.transpose(0, 3, 1, 2), dev)
    for ahead in (1, 2, 4):
        for rows, cpw in ((-1, -1), (1, 1), (2, 3), (3, 2), (5, 4), (7, 1), (64, 1)):
            for grp in (0, 1, 2):
                with _lib.debug_override(u8_ahead=ahead, remap_rows=rows, remap_cpw=cpw, remap_noswz=grp):
                    hwc = N(cu.remap_separable(ti, tx, ty, mode="cv2", channels_last=True))
                    chw = N(cu.remap_separable(tc, tx, ty, mode="cv2"))
                assert np.array_equal(hwc, ref), (ahead, rows, cpw, grp, "hwc")
                assert np.array_equal(chw.transpose(0, 2, 3, 1), ref), (ahead, rows, cpw, grp, "chw")


@pytest.mark.parametrize("C", [1, 2, 3, 4])
@pytest.mark.parametrize("kind", ["cdf", "wild", "identity"])
def test_remap_cv2_staged_channels_and_edges(dev, C, kind):
    """cv2 arithmetic on the staged kernels for every channel count, with coordinates on the 1/64-pixel rounding
    ties of cvRound (half to even), exactly on pixel centres, and far outside the image."""
    from attwarp_amd import checkpoint_utils as cu
    rng = np.random.default_rng(900 + C)
    B, H, W, Ho, Wo = 2, 48, 64, 52, 72
    mx, my = make_maps(rng, B, H, W, Ho, Wo, kind)
    mx[:, :8] = np.array([0.015625, 0.046875, 3.0, 62.984375, 63.0, 63.5, -7.25, 1e9], np.float32)   # ties, edges, far
    my[:, :6] = np.array([0.015625, 47.0, 46.984375, 47.515625, -1e9, 5.5], np.float32)
    for dt in (np.float32, np.uint8):
        img = rng.random((B, H, W, C), dtype=np.float32)
        if dt == np.uint8:
            img = rng.integers(0, 256, (B, H, W, C), dtype=np.uint8)
            img[0, :4, :4] = 255; img[0, 4:8, :4] = 0                       # saturated corners
        ref = np.stack([O.remap_bilinear(img[b], mx[b], my[b], "cv2") for b in range(B)])
        hwc = N(cu.remap_separable(T(img, dev), T(mx, dev), T(my, dev), mode="cv2", channels_last=True))
        chw = N(cu.remap_separable(T(img.transpose(0, 3, 1, 2), dev), T(mx, dev), T(my, dev), mode="cv2"))
        assert np.array_equal(hwc, ref), (dt.__name__, "hwc")
        assert np.array_equal(chw.transpose(0, 2, 3, 1), ref), (dt.__name__, "chw")


@pytest.mark.parametrize("kind", ["cdf", "wild"])
@pytest.mark.parametrize("split", [-1, 0, 1])
@pytest.mark.parametrize("mode", ["exact", "cv2"])
def test_remap_planar_plane_split(dev, kind, split, mode):
    """Planar float32 images with wide rows are dispatched plane by plane (maps of image b serve planes b*C..):
    default heuristic, forced on and forced off must all equal the oracle bit-for-bit."""
    from attwarp_amd import checkpoint_utils as cu
    rng = np.random.default_rng(77)
    B, C, H, W, Ho, Wo = 3, 3, 40, 1024, 37, 768
    img = rng.random((B, C, H, W), dtype=np.float32)
    mx, my = make_maps(rng, B, H, W, Ho, Wo, kind)
    ref = np.stack([O.remap_bilinear(img[b].transpose(1, 2, 0), mx[b], my[b], mode) for b in range(B)]).transpose(0, 3, 1, 2)
    with _lib.debug_override(remap_chw_split=split):
        got = N(cu.remap_separable(T(img, dev), T(mx, dev), T(my, dev), mode=mode))
        two = N(cu.remap_separable(T(img[:, :2], dev), T(mx, dev), T(my, dev), mode=mode))      # C = 2
    assert np.array_equal(got, ref)
    assert np.array_equal(two, ref[:, :2])


@pytest.mark.parametrize("shape", [
    (2, 37, 1500, 33, 1400, 3, "hwc"),     # rows of 4500 / 4200 floats: two column tiles
    (2, 29, 1368, 40, 2900, 3, "hwc"),     # strong magnification: three tiles, narrow source spans
    (1, 33, 5000, 31, 4200, 1, "chw"),     # single plane wider than 4096
    (2, 30, 4400, 26, 1400, 3, "chw"),     # planar, split into one-channel images wider than 4096 -> tiled
    (2, 26, 1100, 26, 1100, 4, "hwc"),     # 4 channels, 4400 floats
])
@pytest.mark.parametrize("kind", ["cdf", "wild", "identity"])
@pytest.mark.parametrize("mode", ["exact", "cv2"])
def test_remap_wide_rows_column_tiles(dev, shape, kind, mode):
    """Rows wider than the 4096-float LDS row run in column tiles ("cdf": every tile staged; "wild": spans that do not
    fit take the per-tile direct path); both must equal the oracle and the generic kernel bit-for-bit."""
    from attwarp_amd import checkpoint_utils as cu
    B, H, W, Ho, Wo, C, layout = shape
    rng = np.random.default_rng(hash((shape, kind)) % 2**32)
    img = rng.random((B, H, W, C), dtype=np.float32)
    mx, my = make_maps(rng, B, H, W, Ho, Wo, kind)
    ref = np.stack([O.remap_bilinear(img[b], mx[b], my[b], mode) for b in range(B)])
    x = T(img if layout == "hwc" else img.transpose(0, 3, 1, 2), dev)
    got = N(cu.remap_separable(x, T(mx, dev), T(my, dev), mode=mode, channels_last=(layout == "hwc")))
    with _lib.debug_override(remap_tiled=0):          # generic gather kernel
        gen = N(cu.remap_separable(x, T(mx, dev), T(my, dev), mode=mode, channels_last=(layout == "hwc")))
    if mode == "exact":
        with _lib.debug_override(remap_tile_ko=12):   # the wider tile variant
            t12 = N(cu.remap_separable(x, T(mx, dev), T(my, dev), mode=mode, channels_last=(layout == "hwc")))
        assert np.array_equal(t12, got)
    if layout == "chw":
        got, gen = got.transpose(0, 2, 3, 1), gen.transpose(0, 2, 3, 1)
    assert np.array_equal(got, ref) and np.array_equal(gen, ref)
    # the uint8 kernel tiles rows wider than 4096 BYTES: same shapes, 4x the width in bytes is not needed --
    # these rows (4200 .. 5000 bytes) already exceed it
    img8 = (img * 255).astype(np.uint8)
    ref8 = np.stack([O.remap_bilinear(img8[b], mx[b], my[b], mode) for b in range(B)])
    x8 = T(img8 if layout == "hwc" else img8.transpose(0, 3, 1, 2), dev)
    got8 = N(cu.remap_separable(x8, T(mx, dev), T(my, dev), mode=mode, channels_last=(layout == "hwc")))
    if layout == "chw":
        got8 = got8.transpose(0, 2, 3, 1)
    assert np.array_equal(got8, ref8)


@pytest.mark.parametrize("shape", [(40, 56, 33, 47, 3, "hwc"), (336, 336, 336, 336, 3, "hwc"), (64, 1024, 50, 1024, 3, "chw"),
                                   (30, 1100, 26, 1100, 4, "hwc"), (31, 29, 31, 29, 2, "chw")])
@pytest.mark.parametrize("mode", ["exact", "cv2"])
def test_remap_non_finite_and_huge_coordinates(dev, shape, mode):
    """NaN, +-Inf, +-1e30, +-3e9, the int32 edge of cvRound(32 m) and other out-of-range coordinates sprinkled over
    both maps: every kernel family (generic, staged rows, plane split, column tiles; float32 and uint8, integer uint8
    form) follows the oracle's conventions bit for bit (cv2: x86 cvRound -> INT_MIN -> pixel 0; exact: clamp to
    [-1, size], NaN = -1) and no output is NaN.  (tests/fuzz/fuzz_remap.py is the randomised long form of this test.)"""
    from attwarp_amd import checkpoint_utils as cu
    H, W, Ho, Wo, C, layout = shape
    rng = np.random.default_rng(H * 7 + Wo)
    B = 2
    mx = (rng.random((B, Wo)) * W).astype(np.float32)
    my = (rng.random((B, Ho)) * H).astype(np.float32)
    sp = np.array([np.nan, np.inf, -np.inf, 1e30, -1e30, 3e9, -3e9, 6.8e7, 67108860.0, -67108864.0, -67108868.0, 65536.0,
                   -65536.0, W - 1, W - 0.5, -0.5, -1.0, -1.5, W + 0.25, 1 / 64, 3 / 64], np.float32)
    for m in (mx, my):
        k = max(4, m.size // 6)
        m.reshape(-1)[rng.integers(0, m.size, k)] = rng.choice(sp, k)
    mx[0, :len(sp)] = sp[:Wo][:len(sp)] if Wo >= len(sp) else mx[0, :len(sp)]
    for dt in (np.float32, np.uint8):
        img = rng.random((B, H, W, C), dtype=np.float32)
        if dt == np.uint8:
            img = (img * 255).astype(np.uint8)
        with np.errstate(all="ignore"):
            ref = np.stack([O.remap_bilinear(img[b], mx[b], my[b], mode) for b in range(B)])
        assert np.isfinite(ref.astype(np.float64)).all()
        x = T(img if layout == "hwc" else np.ascontiguousarray(img.transpose(0, 3, 1, 2)), dev)
        for over in ({}, {"remap_variant": 1}, {"remap_variant": 2}):
            with _lib.debug_override(**over):
                got = N(cu.remap_separable(x, T(mx, dev), T(my, dev), mode=mode, channels_last=(layout == "hwc")))
            if layout == "chw":
                got = got.transpose(0, 2, 3, 1)
            assert np.array_equal(got, ref), (shape, mode, dt.__name__, over)


@pytest.mark.parametrize("R", [1, 5, 64])
@pytest.mark.parametrize("mode", ["exact", "cv2"])
def test_remap_rows_block_boundaries(dev, R, mode):
    """Row-block size must not change a single bit (halo / slide logic at block seams)."""
    from attwarp_amd import checkpoint_utils as cu
    rng = np.random.default_rng(41)
    H = W = 96
    img = rng.random((2, H, W, 3), dtype=np.float32)
    px = softmax_rows(rng, (2, 24)) ** 3; px /= px.sum(1, keepdims=True)      # strongly peaked: slopes >> 2 and << 1
    Fx = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(px.astype(np.float32), W), 0))
    mx, my = O.maps_from_cdf(Fx, Fx[::-1].copy(), (H, W))
    ref = np.stack([O.remap_bilinear(img[b], mx[b], my[b], mode) for b in range(2)])
    img8 = (img * 255).astype(np.uint8)
    ref8 = np.stack([O.remap_bilinear(img8[b], mx[b], my[b], mode) for b in range(2)])
    with _lib.debug_override(remap_rows=R):
        got = N(cu.remap_separable(T(img, dev), T(mx, dev), T(my, dev), mode=mode, channels_last=True))
        got8 = N(cu.remap_separable(T(img8, dev), T(mx, dev), T(my, dev), mode=mode, channels_last=True))
    assert np.array_equal(got, ref)
    assert np.array_equal(got8, ref8)
    # nontemporal loads and stores (measurement options of the tuning flavour) change no bit
    with _lib.debug_override(remap_rows=R, remap_nt=3):
        got_nt = N(cu.remap_separable(T(img, dev), T(mx, dev), T(my, dev), mode=mode, channels_last=True))
        chw_nt = N(cu.remap_separable(T(img.transpose(0, 3, 1, 2), dev), T(mx, dev), T(my, dev), mode=mode))
    assert np.array_equal(got_nt, ref) and np.array_equal(chw_nt.transpose(0, 2, 3, 1), ref)


@pytest.mark.parametrize("dt", [np.float32, np.uint8])
def test_remap_cv2_compat_mode(dev, dt):
    from attwarp_amd import checkpoint_utils as cu
    rng = np.random.default_rng(42)
    B, H, W, C = 2, 40, 56, 3
    img = rng.random((B, H, W, C), dtype=np.float32)
    if dt == np.uint8:
        img = (img * 255).astype(np.uint8)
    mx, my = make_maps(rng, B, H, W, 64, 48, "cdf")
    ref = np.stack([O.remap_bilinear(img[b], mx[b], my[b], "cv2") for b in range(B)])
    got = N(cu.remap_separable(T(img, dev), T(mx, dev), T(my, dev), mode="cv2", channels_last=True))
    assert np.array_equal(got, ref)
    got_chw = N(cu.remap_separable(T(img.transpose(0, 3, 1, 2), dev), T(mx, dev), T(my, dev), mode="cv2"))
    assert np.array_equal(got_chw.transpose(0, 2, 3, 1), ref)


def test_remap_rejects_bad_input(dev):
    from attwarp_amd import checkpoint_utils as cu, _lib
    img = torch.zeros(1, 5, 8, 8, device=dev)
    m = torch.zeros(1, 8, device=dev)
    with pytest.raises(_lib.AttWarpError, match="C=5"):
        cu.remap_separable(img, m, m)
    with pytest.raises(TypeError):
        cu.remap_separable(img.half(), m, m)          # float16: not a cv2.remap depth either (float64 is: pass-through)
    with pytest.raises(KeyError):
        cu.remap_separable(img[:, :3], m, m, mode="nearest")


# =============================== composites ==================================
@pytest.mark.parametrize("dtype", [torch.float32, torch.uint8, torch.float16])
def test_warp_from_cdf_torch_surface(dev, dtype):
    from attwarp_amd import checkpoint_utils as cu
    rng = np.random.default_rng(51)
    B, C, H, W = 3, 3, 72, 88
    img = rng.random((B, C, H, W), dtype=np.float32)
    px = softmax_rows(rng, (B, 24)); py = softmax_rows(rng, (B, 24))
    Fx = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(px, W), 0))
    Fy = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(py, H), 0))
    if dtype == torch.uint8:
        src = (img * 255).astype(np.uint8)
        t_img = T(src, dev)
    else:
        t_img = T(img, dev).to(dtype)
        src = N(t_img.float())
    for out_size in (None, (50, 120)):
        out = cu.warp_from_cdf_torch(t_img, T(Fx, dev), T(Fy, dev), out_size)
        assert out.dtype == dtype and out.device == t_img.device          # reference contract :203
        ref = O.warp_from_cdf(src, Fx, Fy, out_size)
        if dtype == torch.float16:
            assert np.array_equal(N(out.float()), ref.astype(np.float16).astype(np.float32))
        else:
            assert np.array_equal(N(out), ref)


def test_warp_image_by_attention_and_save(dev, golden, tmp_path):
    from PIL import Image
    from attwarp_amd import new_method as nm
    g = golden("maps_from_attention")
    att = g["att_u8"]                                   # 336x336 uint8 LANCZOS-upsampled mask
    rng = np.random.default_rng(52)
    img = rng.integers(0, 256, (336, 336, 3), dtype=np.uint8)
    for tr in ("identity", "sqrt", "square"):
        got = nm.warp_image_by_attention(img, att, 500, 500, transform=tr)
        assert got.shape == (500, 500, 3) and got.dtype == np.uint8
        ref = O.warp_image_by_attention(img, att, 500, 500, tr)
        assert np.array_equal(got, ref), tr                   # maps are bit-exact, so are the pixels
    # module-state form (reference behaviour) == explicit form
    nm.set_transform_function("sqrt")
    assert np.array_equal(nm.warp_image_by_attention(img, att, 500, 500),
                          nm.warp_image_by_attention(img, att, 500, 500, transform="sqrt"))
    # float32 image, grayscale image
    imgf = rng.random((336, 336, 3), dtype=np.float32)
    gotf = nm.warp_image_by_attention(imgf, att, 336, 336, transform="identity")
    assert np.abs(gotf - O.warp_image_by_attention(imgf, att, 336, 336, "identity")).max() <= TOL_PIXEL
    gray = nm.warp_image_by_attention(img[:, :, 0], att, 100, 120, transform="identity")
    assert gray.shape == (120, 100)
    # save_warped_image: PIL image in (RGB), file out, True; same pixels as the in-memory warp
    out_path = tmp_path / "w.png"
    ok = nm.save_warped_image(Image.fromarray(img), att, str(tmp_path / "orig.png"), None, str(out_path),
                              width=500, height=500, transform="identity")
    assert ok is True
    saved = np.array(Image.open(out_path))
    expect = nm.warp_image_by_attention(img[:, :, ::-1].copy(), att, 500, 500, transform="identity")[:, :, ::-1]
    assert np.array_equal(saved, expect)
    assert np.array_equal(np.array(Image.open(tmp_path / "orig.png")), img)
    # attention map of another size -> image is resized to it first (reference :478)
    ok = nm.save_warped_image(Image.fromarray(img), np.ones((100, 120), np.float32), None, None, str(out_path))
    assert ok and np.array(Image.open(out_path)).shape == (500, 500, 3)
    # the reference fails where cv2.cvtColor(COLOR_RGB2BGR) fails (new_method.py:421-422 -> :504-506 returns False): a mode-L
    # PIL image is a 2-D array there; an RGBA image is a 4-channel source OpenCV accepts (alpha dropped, 3 channels out)
    gone = tmp_path / "never.png"
    assert nm.save_warped_image(Image.fromarray(img[:, :, 0], mode="L"), att, None, None, str(gone), width=500, height=500) is False
    assert not gone.exists()
    rgba = np.concatenate([img, rng.integers(0, 256, (336, 336, 1), dtype=np.uint8)], axis=2)
    assert nm.save_warped_image(Image.fromarray(rgba, mode="RGBA"), att, None, None, str(out_path), width=500, height=500) is True
    assert np.array_equal(np.array(Image.open(out_path)), expect)


def test_pipeline_attention_stack_end_to_end(dev):
    """Config-2 shape of the path at a small size: stack -> 24x24 -> marginals -> CDF -> maps -> warp."""
    from attwarp_amd import pipeline
    rng = np.random.default_rng(53)
    B, T_, heads, kv, S = 4, 6, 32, 640, 112
    rows = softmax_rows(rng, (T_, B, heads, kv), peak=(250, 300))
    starts = (35 + np.arange(B) % 8).astype(np.int32)
    img = rng.random((B, S, S, 3), dtype=np.float32)
    att = O.attn_reduce_stack(rows, starts).reshape(B, 1, 24, 24)
    px, py = O.gt_marginals(att)
    Fx = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(px, S), 0))
    Fy = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(py, S), 0))
    mx, my = O.maps_from_cdf(Fx, Fy)
    ref = np.stack([O.remap_bilinear(img[b], mx[b], my[b]) for b in range(B)])
    out = pipeline.warp_from_attention_stack(T(img, dev), T(rows, dev), T(starts, dev), channels_last=True)
    assert np.array_equal(N(out), ref)
    # the fused maps launch is bit-identical to the staged one, and returns the aggregated map
    steps = pipeline.attention_step_maps(T(rows, dev), T(starts, dev))
    fmx, fmy, fatt = pipeline.axis_maps_from_attention_steps(steps, (S, S), return_attention=True)
    assert np.array_equal(N(fmx), mx) and np.array_equal(N(fmy), my)
    assert np.array_equal(N(fatt), att.reshape(B, 576))
    fmx2, fmy2 = pipeline.axis_maps_from_attention_steps(steps, (S, S), out_size=(150, 90))
    rx2, ry2 = O.maps_from_cdf(Fx, Fy, (150, 90))
    assert np.array_equal(N(fmx2), rx2) and np.array_equal(N(fmy2), ry2)
    # fp16 attention: the same three launches, A2 rounded in the model dtype (vs the oracle chain A1+A2 in fp16 -> float
    # -> marginals -> maps)
    out16 = pipeline.warp_from_attention_stack(T(img, dev), T(rows, dev).half(), T(starts, dev), channels_last=True)
    att16 = O.attn_reduce_stack(rows.astype(np.float16), starts).astype(np.float32).reshape(B, 1, 24, 24)
    p16x, p16y = O.gt_marginals(att16)
    F16x = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(p16x, S), 0))
    F16y = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(p16y, S), 0))
    m16x, m16y = O.maps_from_cdf(F16x, F16y)
    assert np.array_equal(N(out16), np.stack([O.remap_bilinear(img[b], m16x[b], m16y[b]) for b in range(B)]))
    out_chw = pipeline.warp_from_attention_stack(T(img.transpose(0, 3, 1, 2), dev), T(rows, dev), T(starts, dev))
    assert np.array_equal(N(out_chw).transpose(0, 2, 3, 1), ref)
    # HIP-graph replay of the same step gives the same bytes
    ti, tr, ts = T(img, dev), T(rows, dev), T(starts, dev)
    graph, gout = pipeline.capture_step(lambda: pipeline.warp_from_attention_stack(ti, tr, ts, channels_last=True))
    gout.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(N(gout), ref)


def test_pipeline_masks_chain_main_batched(dev, golden):
    """main_batched.py:243-287: revise_mask -> uint8 -> LANCZOS -> float64 marginals -> 500x500 warp."""
    from attwarp_amd import pipeline
    g = golden("mask_postproc")
    rng = np.random.default_rng(54)
    B, S = 4, 336
    imgs = rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8)
    out = N(pipeline.warp_from_masks(T(imgs, dev), T(g["masks"], dev), (500, 500)))
    assert out.shape == (B, 500, 500, 3)
    bad = 0
    for b in range(B):
        mota = O.lanczos_resize_u8(O.mask_to_u8(O.revise_mask(g["masks"][b], 3, 10)), S, S)
        ref = O.warp_image_by_attention(imgs[b], mota, 500, 500, "identity")
        d = np.abs(out[b].astype(int) - ref.astype(int))
        bad += (d > 0).sum()
        assert d.max() <= 1
    assert bad <= 1e-4 * out.size


@pytest.mark.parametrize("cfg", [(64, 336, 500), (256, 1024, 500)])
@pytest.mark.parametrize("form", ["serial", "stream"])
def test_main_batched_chain_full_size_vs_oracle(dev, cfg, form):
    """The reference's own chain (AGW/main_batched.py:243-287) at full size -- EVERY image of B=64 @ 336 -> 500 and of
    B=256 @ 1024 -> 500 -- against the oracle STAGE BY STAGE: the revised mask (<= 1 ulp), the uint8
    LANCZOS mask given the GPU's revised mask (bit-exact; from the oracle's own revised mask at most a few cells flip by
    1 LSB where the x255 truncation sits on a 1-ulp difference), the float32 maps given the mask (bit-exact), the uint8
    pixels given the maps (bit-exact), for pipeline.warp_from_masks and for the one-launch stream step
    (pipeline.MaskChainStream), whose outputs must equal the serial ones for ALL images."""
    import subprocess
    from attwarp_amd import pipeline, attention_extraction as ae, new_method as nm
    from oracle import c_oracle as C
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "oracle")], check=True, capture_output=True)
    B, S, So = cfg
    g = torch.Generator(device=dev).manual_seed(300 + S)
    imgs = torch.randint(0, 256, (B, S, S, 3), device=dev, dtype=torch.uint8, generator=g)
    att = torch.rand(B, 24, 24, device=dev, generator=g) ** 3
    att = att / att.sum((1, 2), keepdim=True)
    out = pipeline.warp_from_masks(imgs, att, (So, So))
    if form == "stream":
        ring_i = [imgs, imgs.flip(0).contiguous()] * 3
        ring_m = [att, att.flip(0).contiguous()] * 3
        mc = pipeline.MaskChainStream(ring_i, ring_m, (So, So))
        assert mc.pattern == "fused"
        mc.prime(); mc.run(2); mc.drain()
        assert torch.equal(mc.outs[0], out) and torch.equal(mc.outs[4], out)
        assert torch.equal(mc.outs[1], pipeline.warp_from_masks(ring_i[1], ring_m[1], (So, So)))
        return
    rev = ae.revise_mask(att)
    mota = ae.upsample_mask_lanczos(rev, (S, S))
    mx, my = nm.attention_axis_maps(mota, So, So, "identity")
    assert torch.equal(nm.remap_hwc(imgs, mx, my, "cv2"), out)           # the drop-in is these four launches
    att_h, rev_h, mx_h, my_h = N(att), N(rev), N(mx), N(my)
    flips = [0] * B
    for c0 in range(0, B, 32):                                    # every image, 32 at a time through host memory
        sel = list(range(c0, min(c0 + 32, B)))
        mota_h, img_h, out_h = N(mota[c0:c0 + 32]), N(imgs[c0:c0 + 32]), N(out[c0:c0 + 32])
        def check(b):
            i = b - c0
            orev = O.revise_mask(att_h[b], 3, 10)
            u = np.abs(rev_h[b].view(np.int32).astype(np.int64) - orev.view(np.int32).astype(np.int64))
            assert u.max() <= 1, b
            # given the GPU's revised mask everything downstream is bit-exact
            assert np.array_equal(mota_h[i], O.lanczos_resize_u8(O.mask_to_u8(rev_h[b]), S, S)), b
            omx, omy = O.maps_from_attention(mota_h[i], So, So, "identity")
            assert np.array_equal(mx_h[b], omx) and np.array_equal(my_h[b], omy), b
            assert np.array_equal(out_h[i], C.remap_bilinear_u8(img_h[i], omx, omy, "cv2")), b
            # from the oracle's own revised mask: 24 x 24 cells may flip by one grey level, nothing else
            d = np.abs(O.mask_to_u8(orev).astype(int) - O.mask_to_u8(rev_h[b]).astype(int))
            assert d.max() <= 1
            flips[b] = int((d > 0).sum())
        for_each(sel, check)
    flips, sel = sum(flips), range(B)
    assert flips <= 1e-3 * len(sel) * 576


def test_config1_single_image_chain(dev, golden):
    """BASELINE configs[0] on the GPU: the reference's single-image case (main.py): attention map -> mask ->
    LANCZOS -> float64 marginals -> maps -> uint8 warp, against the maps captured from the reference and the
    oracle's resample."""
    from attwarp_amd import new_method as nm, pipeline, attention_extraction as ae
    g = golden("config1")
    img, att = config1_inputs()
    bgr = img[..., ::-1].copy()
    mota = N(ae.upsample_mask_lanczos(ae.revise_mask(T(att, dev)), (336, 336)))
    d = np.abs(mota.astype(np.int32) - g["mota"].astype(np.int32))
    assert d.max() <= 1 and (d != 0).mean() < 1e-3
    for n in (336, 500):
        mx, my = nm.attention_axis_maps(T(g["mota"][None], dev), n, n, "identity")
        assert np.array_equal(N(mx)[0], g[f"mx_{n}"]) and np.array_equal(N(my)[0], g[f"my_{n}"])     # == reference
        nm.set_transform_function("identity", 1.0, 1.0, False)
        out = nm.warp_image_by_attention(bgr, g["mota"], n, n)                                        # numpy in -> numpy out
        assert isinstance(out, np.ndarray) and out.dtype == np.uint8
        assert np.array_equal(out, O.remap_bilinear(bgr, g[f"mx_{n}"], g[f"my_{n}"]))
    # the batched launcher with the same single image: only the <= 1-grey-level mask cells can differ
    w = N(pipeline.warp_from_masks(T(bgr[None], dev), T(att[None], dev), (500, 500)))[0]
    ref = O.remap_bilinear(bgr, g["mx_500"], g["my_500"])
    assert (w != ref).mean() < 2e-3 and np.abs(w.astype(np.int32) - ref.astype(np.int32)).max() <= 2


def test_run_to_run_determinism(dev):
    """Every reduction has a fixed order (no atomics anywhere): repeated launches and launches on a side stream give
    bit-identical results for the whole chain, including the probe and the uint8 / CLIP kernels."""
    from attwarp_amd import pipeline, attention_extraction as ae
    gen = torch.Generator(device=dev).manual_seed(5)
    B, S = 8, 336
    img = torch.rand(B, S, S, 3, device=dev, generator=gen)
    rows = torch.softmax(torch.randn(6, B, 32, 640, device=dev, generator=gen), -1)
    starts = (35 + torch.arange(B, device=dev) % 8).int()
    img8 = (img * 255).to(torch.uint8)
    m24 = torch.rand(B, 24, 24, device=dev, generator=gen)
    q = torch.randn(B, 32, 128, device=dev, generator=gen).half()
    k = torch.randn(B, 32, 640, 128, device=dev, generator=gen).half()

    def run():
        a = pipeline.warp_from_attention_stack(img, rows, starts, channels_last=True)
        w = pipeline.warp_from_masks(img8, m24, (500, 500))
        c = pipeline.clip_preprocess(w)
        p = ae.probe_last_query(q, k, starts, 576)
        return a.clone(), w.clone(), c.clone(), p.clone()

    first = run()
    for _ in range(3):
        again = run()
        assert all(torch.equal(x, y) for x, y in zip(first, again))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        other = run()
    side.synchronize()
    assert all(torch.equal(x, y) for x, y in zip(first, other))


def test_two_host_threads_two_streams(dev):
    """The library keeps no global mutable state (include/attwarp.h): two host threads driving their own streams
    concurrently produce what each would produce alone."""
    import threading
    from attwarp_amd import pipeline
    gen = torch.Generator(device=dev).manual_seed(9)
    jobs = []
    for t in range(2):
        B, S = 4, 128 + 64 * t
        img = torch.rand(B, 3, S, S, device=dev, generator=gen)
        px = torch.softmax(torch.randn(B, 24, device=dev, generator=gen) * (1 + t), 1)
        py = torch.softmax(torch.randn(B, 24, device=dev, generator=gen), 1)
        jobs.append((img, px, py, pipeline.warp_from_pdf(img, px, py).clone()))
    torch.cuda.synchronize()
    errors = []

    def worker(i):
        try:
            img, px, py, ref = jobs[i]
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                for _ in range(30):
                    out = pipeline.warp_from_pdf(img, px, py)
                stream.synchronize()
                if not torch.equal(out, ref):
                    errors.append(f"thread {i}: result differs")
        except Exception as e:          # noqa: BLE001 - report through the list, the assert below fails the test
            errors.append(f"thread {i}: {e!r}")

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


# =============================== full-size properties =========================
@pytest.mark.parametrize("cfg", [(64, 336), (256, 1024)])
@pytest.mark.parametrize("mode", ["exact", "cv2"])
def test_full_size_properties(dev, cfg, mode):
    """BASELINE configs 2 and 3 at full size, both arithmetic modes, checked on the GPU through properties:
    (1) identity maps reproduce the input bit-for-bit; (2) the streaming kernel and the independent
    gather kernel agree bit-for-bit on attention-driven maps (HWC and CHW, float32 and uint8); (3) a constant image
    stays constant (partition of unity; in cv2 mode the four table weights sum to 1 exactly); (4) every output lies
    within the input's range; (5) first and last image equal the CPU oracle bit-for-bit (every image:
    test_whole_batch_vs_c_oracle)."""
    from attwarp_amd import checkpoint_utils as cu, pipeline
    B, S = cfg
    gen = torch.Generator(device=dev).manual_seed(0)
    img = torch.rand((B, S, S, 3), device=dev, generator=gen)
    ar = torch.arange(S, device=dev, dtype=torch.float32).repeat(B, 1)
    out = cu.remap_separable(img, ar, ar, mode=mode, channels_last=True)
    assert torch.equal(out, img)
    px = torch.softmax(torch.randn(B, 24, device=dev, generator=gen) * 2, 1)
    py = torch.softmax(torch.randn(B, 24, device=dev, generator=gen) * 2, 1)
    mx, my = pipeline.axis_maps_from_pdf(px, py, (S, S))
    assert bool((mx[:, 1:] >= mx[:, :-1]).all()) and float(mx.min()) >= 0 and float(mx.max()) <= S
    del out
    a = cu.remap_separable(img, mx, my, mode=mode, channels_last=True)
    with _lib.debug_override(remap_variant=1):     # the independent gather kernel
        b = cu.remap_separable(img, mx, my, mode=mode, channels_last=True)
    assert torch.equal(a, b)
    assert float(a.min()) >= float(img.min()) and float(a.max()) <= float(img.max())
    del b
    const = torch.full_like(img, 0.3125)
    c = cu.remap_separable(const, mx, my, mode=mode, channels_last=True)
    assert torch.equal(c, const)
    del c, const
    # spot-check a few images against the CPU oracle
    for bi in (0, B - 1):
        ref = O.remap_bilinear(N(img[bi]), N(mx[bi]), N(my[bi]), mode)
        assert np.array_equal(N(a[bi]), ref)
    # planar layout: same pixels
    chw = cu.remap_separable(img.permute(0, 3, 1, 2).contiguous(), mx, my, mode=mode)
    assert torch.equal(chw.permute(0, 2, 3, 1), a)
    del chw, a
    # uint8 (the main_batched dtype): rows kernel == gather kernel == oracle
    img8 = (img * 255).to(torch.uint8)
    a8 = cu.remap_separable(img8, mx, my, mode=mode, channels_last=True)
    with _lib.debug_override(remap_variant=1):
        b8 = cu.remap_separable(img8, mx, my, mode=mode, channels_last=True)
    assert torch.equal(a8, b8)
    for bi in (0, B - 1):
        assert np.array_equal(N(a8[bi]), O.remap_bilinear(N(img8[bi]), N(mx[bi]), N(my[bi]), mode))
    c8 = cu.remap_separable(img8.permute(0, 3, 1, 2).contiguous(), mx, my, mode=mode)
    assert torch.equal(c8.permute(0, 2, 3, 1), a8)


def _attention24(B, kind, dev, gen):
    """[B,24,24] attention maps of SURVEY 8d's value distributions."""
    if kind == "random":
        return torch.softmax(torch.randn(B, 576, device=dev, generator=gen) * 2, 1).view(B, 24, 24)
    if kind == "zero":                                 # the fallback branch (AGW/new_method.py:231-239 / clamp_min(1e-6))
        return torch.zeros(B, 24, 24, device=dev)
    att = torch.full((B, 24, 24), 1.0, device=dev)     # peaked: one 3x3 hot spot x100
    cy = torch.randint(1, 23, (B,), device=dev, generator=gen).tolist()
    cx = torch.randint(1, 23, (B,), device=dev, generator=gen).tolist()
    for b in range(B):
        att[b, cy[b] - 1:cy[b] + 2, cx[b] - 1:cx[b] + 2] = 100.0
    return att / att.sum((1, 2), keepdim=True)


@pytest.mark.parametrize("cfg", [(64, 336), (256, 1024)])
@pytest.mark.parametrize("mode", ["cv2", "exact"])
@pytest.mark.parametrize("kind", ["random", "peaked", "zero", "smooth"])
def test_whole_batch_vs_c_oracle(dev, cfg, mode, kind):
    """BASELINE configs[1] and [2] at full size against the ORACLE (oracle/warp_ref.c, itself pinned to the numpy oracle
    and through it to the reference fixtures) -- not against a sibling kernel: EVERY image of B=64 @ 336x336 and of
    B=256 @ 1024x1024, both arithmetic modes, HWC and CHW, float32 and uint8,
    maps and pixels bit for bit; on SURVEY 8d's value distributions: random-softmax attention on white noise, one 3x3 hot
    spot x100 (strong magnification / minification), all-zero attention (the fallback branch), and a smooth image."""
    import subprocess
    from attwarp_amd import checkpoint_utils as cu, pipeline
    from oracle import c_oracle as C
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "oracle")], check=True, capture_output=True)
    B, S = cfg
    gen = torch.Generator(device=dev).manual_seed(17 + S)
    if kind == "smooth":
        yy = torch.linspace(0, 1, S, device=dev)[None, :, None, None]
        xx = torch.linspace(0, 1, S, device=dev)[None, None, :, None]
        ph = torch.rand((B, 1, 1, 3), device=dev, generator=gen) * 6.28
        img = 0.5 + 0.25 * torch.sin(7.0 * xx + ph) * torch.cos(5.0 * yy - ph) + 0.2 * xx * yy
        att = _attention24(B, "random", dev, gen)
    else:
        img = torch.rand((B, S, S, 3), device=dev, generator=gen)
        att = _attention24(B, kind, dev, gen)
    img = img.contiguous()
    px, py = cu.gt_marginals(att.view(B, 1, 24, 24))
    mx, my = pipeline.axis_maps_from_pdf(px, py, (S, S))
    # maps of EVERY image: the C oracle's own chain from the 24 x 24 map
    inv = O.right_inverse_core(24, S)
    att_h, mx_h, my_h = N(att), N(mx), N(my)
    def check_maps(b):
        opx, opy = C.marginals(att_h[b])
        omx = C.axis_map_from_cdf(C.cdf_from_density(C.right_inverse(opx, S, inv, clamp0=True)), S)
        omy = C.axis_map_from_cdf(C.cdf_from_density(C.right_inverse(opy, S, inv, clamp0=True)), S)
        assert np.array_equal(mx_h[b], omx) and np.array_equal(my_h[b], omy), b
    for_each(range(B), check_maps)
    if kind == "zero":
        assert float(mx.max()) <= S and bool(torch.isfinite(mx).all())
    img8 = (img * 255).to(torch.uint8)
    a = cu.remap_separable(img, mx, my, mode=mode, channels_last=True)
    a8 = cu.remap_separable(img8, mx, my, mode=mode, channels_last=True)
    c = cu.remap_separable(img.permute(0, 3, 1, 2).contiguous(), mx, my, mode=mode)
    c8 = cu.remap_separable(img8.permute(0, 3, 1, 2).contiguous(), mx, my, mode=mode)
    assert torch.equal(c.permute(0, 2, 3, 1), a) and torch.equal(c8.permute(0, 2, 3, 1), a8)     # all B images
    for c0 in range(0, B, 32):                                    # every image, 32 at a time through host memory
        img_h, img8_h, a_h, a8_h = N(img[c0:c0 + 32]), N(img8[c0:c0 + 32]), N(a[c0:c0 + 32]), N(a8[c0:c0 + 32])
        def check(b):
            i = b - c0
            assert np.array_equal(a_h[i], C.remap_bilinear(img_h[i], mx_h[b], my_h[b], "hwc", mode)), (b, "float32")
            assert np.array_equal(a8_h[i], C.remap_bilinear_u8(img8_h[i], mx_h[b], my_h[b], mode)), (b, "uint8")
        for_each(range(c0, min(c0 + 32, B)), check)


@pytest.mark.parametrize("cfg", [(64, 336), (256, 1024)])
@pytest.mark.parametrize("adt", [torch.float32, torch.float16])
def test_stack_chain_full_size_vs_oracle(dev, cfg, adt):
    """The exact chain bench.py times, at its full sizes, FROM THE STACK: attention rows [T=20, B, 32 heads, kv=640]
    (float32, and float16 as LLaVA emits them), image tokens at 35 + b mod 8 -> A1 -> A2 .. A11 -> A12 on [B,S,S,3] float32.
    The aggregated 24 x 24 maps, the 1-D maps and the pixels of EVERY image (both sizes) against the
    oracle chain (numpy A1/A2 in the row dtype, oracle/warp_ref.c for everything behind it; float32 rows also through
    warp_ref.c's own whole-path entry point).  The stream forms (pipeline.OverlappedWarp patterns "am" and "fused") must
    produce the same bytes as the serial launches for ALL images."""
    import subprocess
    from attwarp_amd import pipeline
    from oracle import c_oracle as C
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "oracle")], check=True, capture_output=True)
    B, S = cfg
    T_, heads, kv = 20, 32, 640
    gen = torch.Generator(device=dev).manual_seed(1234 + S)
    img = torch.rand((B, S, S, 3), device=dev, generator=gen)
    rows = torch.empty((T_, B, heads, kv), device=dev)
    for t in range(T_):
        rows[t] = torch.softmax(torch.randn((B, heads, kv), device=dev, generator=gen), dim=-1)
    rows = rows.to(adt)
    starts = (35 + torch.arange(B, device=dev) % 8).to(torch.int32)
    steps = pipeline.attention_step_maps(rows, starts)
    mx, my, att = pipeline.axis_maps_from_attention_steps(steps, (S, S), return_attention=True)
    out = pipeline.warp_from_attention_stack(img, rows, starts, channels_last=True, mode="cv2")
    st_h, att_h, mx_h, my_h = N(starts), N(att), N(mx), N(my)
    inv = O.right_inverse_core(24, S)
    for c0 in range(0, B, 32):                                    # every image, 32 at a time through host memory
        rows_h = N(rows[:, c0:c0 + 32])
        att_o = O.attn_reduce_stack(rows_h, st_h[c0:c0 + 32])     # numpy A1 + A2 in the row dtype
        assert att_o.dtype == (np.float32 if adt == torch.float32 else np.float16)
        assert np.array_equal(att_h[c0:c0 + 32], att_o.astype(np.float32))
        img_h, out_h = N(img[c0:c0 + 32]), N(out[c0:c0 + 32])
        def check(b):
            i = b - c0
            a24 = att_o[i].astype(np.float32).reshape(24, 24)
            opx, opy = C.marginals(a24)
            omx = C.axis_map_from_cdf(C.cdf_from_density(C.right_inverse(opx, S, inv, clamp0=True)), S)
            omy = C.axis_map_from_cdf(C.cdf_from_density(C.right_inverse(opy, S, inv, clamp0=True)), S)
            assert np.array_equal(mx_h[b], omx) and np.array_equal(my_h[b], omy), b
            assert np.array_equal(out_h[i], C.remap_bilinear(img_h[i], omx, omy, "hwc", "cv2")), b
            if adt == torch.float32 and b % 8 == 0:                # the C oracle's own whole path (its A1 included)
                assert np.array_equal(out_h[i], C.warp_from_attention_stack(img_h[i], np.ascontiguousarray(rows_h[:, i]), st_h[b], inv, inv, "hwc", "cv2")), b
        for_each(range(c0, min(c0 + 32, B)), check)
    del img_h, out_h, rows_h
    for pattern in ("am", "fused"):
        ow = pipeline.OverlappedWarp(img, rows, starts, channels_last=True, mode="cv2", pattern=pattern)
        ow.reset(); ow.prime(); ow.prime2()
        got = ow.step()
        assert ow.pattern == pattern and torch.equal(got, out), pattern
        got = ow.step()                                            # second step: the maps built INSIDE the first one
        assert torch.equal(got, out), pattern
        del ow, got


@pytest.mark.parametrize("cfg", [(64, 336), (8, 1024)])
@pytest.mark.parametrize("layout", ["hwc", "chw"])
@pytest.mark.parametrize("kind", ["peaked", "wild"])
def test_remap_exact_vs_torch_grid_sample(dev, cfg, layout, kind):
    """Independent check of the EXACT mode on the GPU box: torch's own F.grid_sample(bilinear, border,
    align_corners=True) -- north_star's named op -- on the same separable maps, at BASELINE image sizes.
    float32 grid_sample: tolerance 1e-4 (north_star) at 336; its own [-1,1] coordinate round trip costs up to
    ~S * 2^-23 pixels (SURVEY 8c measured 2e-5 @336, 2.9e-5 @1024 on average maps; 1.1e-4 on peaked ones), so at
    1024 the float32 comparison allows 2e-4 and the float64 grid_sample below is the tight check (2e-6)."""
    import torch.nn.functional as F
    from attwarp_amd import checkpoint_utils as cu, pipeline
    B, S = cfg
    gen = torch.Generator(device=dev).manual_seed(5)
    img = torch.rand((B, 3, S, S), device=dev, generator=gen)
    if kind == "peaked":
        px = torch.softmax(torch.randn(B, 24, device=dev, generator=gen) * 3, 1)
        py = torch.softmax(torch.randn(B, 24, device=dev, generator=gen) * 3, 1)
        mx, my = pipeline.axis_maps_from_pdf(px, py, (S, S))
    else:
        mx = torch.rand((B, S), device=dev, generator=gen) * (S + 6) - 3
        my = torch.rand((B, S), device=dev, generator=gen) * (S + 6) - 3
    if layout == "hwc":
        got = cu.remap_separable(img.permute(0, 2, 3, 1).contiguous(), mx, my, mode="exact",
                                 channels_last=True).permute(0, 3, 1, 2)
    else:
        got = cu.remap_separable(img, mx, my, mode="exact")
    gx = (2.0 * mx.double() / (S - 1) - 1.0)[:, None, :].expand(B, S, S)
    gy = (2.0 * my.double() / (S - 1) - 1.0)[:, :, None].expand(B, S, S)
    grid = torch.stack([gx, gy], -1).float()
    ref = F.grid_sample(img, grid, mode="bilinear", padding_mode="border", align_corners=True)
    err = float((got - ref).abs().max())
    assert err <= TOL_PIXEL * (1 if S <= 512 else 2), err
    # and in float64 (no coordinate round-trip error left): the kernel's three-rounding lerps stay within 2e-7
    ref64 = F.grid_sample(img.double(), torch.stack([gx, gy], -1), mode="bilinear", padding_mode="border",
                          align_corners=True)
    # grid_sample recomputes the pixel coordinate from the normalised one: allow its float64 round trip
    assert float((got.double() - ref64).abs().max()) <= 2e-6


def test_integration_md_stub_runs(dev):
    """The ctypes stub shown in INTEGRATION.md (what a reference maintainer would paste) must work as written."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(import ctypes, torch.*?)```", text, re.S).group(1)
    cwd = os.getcwd()
    os.chdir(root)
    try:
        ns = {}
        exec(compile(code, "INTEGRATION.md", "exec"), ns)
    finally:
        os.chdir(cwd)
    rng = np.random.default_rng(61)
    img = rng.random((2, 3, 40, 56), dtype=np.float32)
    Fx = O.cdf_from_density(rng.random((2, 56), dtype=np.float32))
    Fy = O.cdf_from_density(rng.random((2, 40), dtype=np.float32))
    out = ns["warp_from_cdf_torch"](T(img, dev), T(Fx, dev), T(Fy, dev), (48, 64))
    torch.cuda.synchronize()
    assert np.array_equal(N(out), O.warp_from_cdf(img, Fx, Fy, (48, 64)))
    with pytest.raises(ValueError):
        ns["warp_from_cdf_torch"](T(img, dev), T(Fx[:, :50], dev), T(Fy, dev))


def test_integration_md_driver_loop_snippet_runs(dev, tmp_path):
    """The replacement INTEGRATION.md shows for the per-sample block of AGW/main_batched.py:243-287, exec'd as written on
    stand-ins for the driver's variables (PIL images of different sizes, 24 x 24 maps): the PNGs it writes hold what the
    reference's chain computes (oracle; `cv2.imwrite` of its BGR result = this RGB array), the saved mota masks are blend_mask's."""
    import re, types
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(from attwarp_amd import pipeline\nimgs = pipeline.upload_images.*?)```", text, re.S).group(1)
    rng = np.random.default_rng(62)
    sizes = [(61, 100), (90, 77), (64, 64)]
    host = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for (h, w) in sizes]
    att = [torch.rand(24, 24, device=dev) ** 2 for _ in sizes]
    wdir, adir = tmp_path / "warped", tmp_path / "att"
    wdir.mkdir(); adir.mkdir()
    ns = dict(b_images=[Image.fromarray(h) for h in host], attn_maps=att, model=types.SimpleNamespace(device=dev), torch=torch, np=np,
              os=os, Image=Image, DEFAULT_HEIGHT=80, DEFAULT_WIDTH=88, enhance_coe=10, kernel_size=3, current_bs=len(sizes),
              sample_id=lambda j: f"img{j}", WARPED_IMAGES_DIR=str(wdir), ATTENTION_MAPS_DIR=str(adir))
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    for j, (h, w) in enumerate(sizes):
        rev = N(__import__("attwarp_amd").attention_extraction.revise_mask(att[j][None]))[0]
        mota = O.lanczos_resize_u8(O.mask_to_u8(rev), w, h)
        assert np.array_equal(np.load(adir / f"img{j}_mota_mask.npy"), mota)
        ref = O.warp_image_by_attention(host[j], mota, 88, 80, "identity")
        assert np.array_equal(np.asarray(Image.open(wdir / f"img{j}_identity.png")), ref)


def test_hook_plumbing_with_dummy_decoder(dev):
    """batch_hook_logger on a stand-in model: the hook is registered on layers[i].self_attn, the patched
    forward forces output_attentions=True for that layer only, and generation steps accumulate."""
    import types
    from attwarp_amd import attention_extraction as ae

    class Attn(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.calls = []

        def forward(self, hidden, output_attentions=False):
            self.calls.append(output_attentions)
            B, q = hidden.shape[0], hidden.shape[1]
            w = torch.softmax(hidden.new_ones(B, 4, q, 640).cumsum(-1) * 0.01, dim=-1) if output_attentions else None
            return hidden, w, None

    class Layer(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.self_attn = Attn()

    model = types.SimpleNamespace(model=types.SimpleNamespace(layers=torch.nn.ModuleList([Layer() for _ in range(3)])),
                                  config=types.SimpleNamespace(output_attentions=True))
    hl = ae.batch_hook_logger(model, dev, layer_index=1)
    assert model.config.output_attentions is False and model.batch_hooklogger is hl
    hl.set_batch_image_token_ranges([10, 12], [586, 588])
    h = torch.zeros(2, 5, 8, device=dev)
    for i, layer in enumerate(model.model.layers):            # prefill
        layer.self_attn(h)
    for i, layer in enumerate(model.model.layers):            # one decode step
        layer.self_attn(h[:, :1])
    assert model.model.layers[1].self_attn.calls == [True, True]          # forced on the hooked layer
    assert model.model.layers[0].self_attn.calls == [False, False]        # untouched elsewhere
    assert len(hl.step_attentions) == 2 and hl.step_attentions[0].shape == (2, 576)
    maps = hl.finalize_batch()
    assert len(maps) == 2 and maps[0].shape == (24, 24)
    assert abs(float(maps[0].sum()) - 1.0) < 1e-5                         # renormalised over the image tokens
    hl.remove_hook_and_unpatch()
    model.model.layers[1].self_attn(h)
    assert model.model.layers[1].self_attn.calls[-1] is False and len(hl.step_attentions) == 2


def test_single_sample_hook_logger_with_dummy_decoder(dev):
    """hook_logger / MaskHookLogger.register_hook / remove_hook / _attention_hook (reference llava.py:74-92,141-187,
    the surface main.py:38,307 and new_method.py:45 import): config.output_attentions is switched on for the whole
    model, the forward hook of layers[i].self_attn consumes output[1], malformed outputs are ignored, the default
    range [1, 577) is used until set_image_token_range, finalize() equals the oracle."""
    import types
    from attwarp_amd import attention_extraction as ae

    class Attn(torch.nn.Module):
        def __init__(self, cfg):
            super().__init__()
            self.cfg, self.mode = cfg, "weights"

        def forward(self, hidden):
            B, q = hidden.shape[0], hidden.shape[1]
            g = torch.Generator(device=hidden.device).manual_seed(q)
            w = torch.softmax(torch.randn(B, 4, q, 640, device=hidden.device, generator=g), dim=-1)
            if self.mode == "weights":
                return hidden, (w if self.cfg.output_attentions else None), None
            if self.mode == "3d":
                return hidden, w[0], None
            return hidden                                          # not a tuple

    cfg = types.SimpleNamespace(output_attentions=False)
    layers = torch.nn.ModuleList([torch.nn.Module() for _ in range(3)])
    for l in layers:
        l.self_attn = Attn(cfg)
    model = types.SimpleNamespace(model=types.SimpleNamespace(layers=layers), config=cfg)
    hl = ae.hook_logger(model, dev, layer_index=2)
    assert isinstance(hl, ae.MaskHookLogger) and model.hooklogger is hl
    assert cfg.output_attentions is True and model._original_output_attentions is False
    assert hl._find_image_token_range(torch.zeros(1, 700, dtype=torch.long)) == (1, 577)
    h = torch.zeros(1, 6, 8, device=dev)
    seen = []
    for q in (6, 1, 1):                                            # prefill + two decode steps
        for i, l in enumerate(layers):
            l.self_attn(h[:, :q])
        g = torch.Generator(device=dev).manual_seed(q)
        seen.append(torch.softmax(torch.randn(1, 4, q, 640, device=dev, generator=g), dim=-1))
    assert len(hl.attns) == 3 and hl.attns[0].shape == (1, 576)
    got = N(hl.finalize())
    ref = np.mean([O.attn_reduce_step(N(w), [1], [577])[0] for w in seen], axis=0)
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=1e-9)
    hl.reinit()
    hl.set_image_token_range(40, 616)
    layers[2].self_attn(h)
    np.testing.assert_allclose(N(hl.finalize()), O.attn_reduce_step(N(seen[0]), [40], [616])[0], rtol=2e-6, atol=1e-9)
    for mode in ("3d", "plain"):                                   # silently ignored, like the reference
        layers[2].self_attn.mode = mode
        layers[2].self_attn(h)
    assert len(hl.attns) == 1
    layers[2].self_attn.mode = "weights"
    hl.register_hook()                                             # re-registering replaces the handle (no double count)
    layers[2].self_attn(h)
    assert len(hl.attns) == 2
    hl.remove_hook()
    layers[2].self_attn(h)
    assert len(hl.attns) == 2 and hl.hook_handle is None
    assert float(ae.MaskHookLogger(model, dev).finalize().sum()) == pytest.approx(1.0)     # empty -> uniform [576]


@pytest.mark.parametrize("rotary_style", ["seq_len (4.36-4.37)", "position_ids (4.38-4.47)"])
def test_probe_legacy_hook_on_transformers_4_37_style_attention(dev, rotary_style):
    """INTEGRATION.md section 2, the recipe for the reference's pinned transformers 4.37.2 (no AttentionInterface):
    ``register_probe_legacy`` = forward hook that re-applies q_proj + rotary to the last token and reads the layer's
    key cache.  Checked on a stand-in module with 4.37.2's LlamaAttention interface (q_proj / k_proj /
    rotary_emb(x, seq_len) / layer_idx / past_key_value.update) against that module's own eager attention weights
    fed through the reference-style hook (``register_hook_and_patch`` + ``_process_attention``): fp16, left padding,
    prefill + two decode steps."""
    import math
    import types
    from attwarp_amd import attention_extraction as ae
    heads, D, hid = 8, 64, 512                       # head_dim a power of 4: /sqrt(D) == *D**-0.5 exactly

    class Rotary(torch.nn.Module):                   # 4.37.2: forward(x, seq_len) -> (cos[:seq_len], sin[:seq_len])
        def forward(self, x, seq_len=None):
            inv = 1.0 / (10000 ** (torch.arange(0, D, 2, device=x.device).float() / D))
            fr = torch.outer(torch.arange(seq_len, device=x.device).float(), inv)
            emb = torch.cat((fr, fr), dim=-1)
            return emb.cos().to(x.dtype), emb.sin().to(x.dtype)

    class RotaryPos(torch.nn.Module):                # 4.38 - 4.47: forward(x, position_ids) -> (cos, sin) [B, seq, D]
        def forward(self, x, position_ids):
            inv = 1.0 / (10000 ** (torch.arange(0, D, 2, device=x.device).float() / D))
            fr = position_ids[:, :, None].float() * inv[None, None, :]
            emb = torch.cat((fr, fr), dim=-1)
            return emb.cos().to(x.dtype), emb.sin().to(x.dtype)

    new_style = rotary_style.startswith("position_ids")

    class Cache:                                     # DynamicCache of 4.36+: key_cache[layer] grows along the kv axis
        def __init__(self, n):
            self.key_cache = [None] * n

        def update(self, k, layer):
            self.key_cache[layer] = k if self.key_cache[layer] is None else torch.cat([self.key_cache[layer], k], dim=2)
            return self.key_cache[layer]

    class Attn(torch.nn.Module):
        def __init__(self, layer_idx):
            super().__init__()
            self.layer_idx, self.num_heads = layer_idx, heads
            self.q_proj = torch.nn.Linear(hid, heads * D, bias=False)
            self.k_proj = torch.nn.Linear(hid, heads * D, bias=False)
            self.rotary_emb = RotaryPos() if new_style else Rotary()

        def forward(self, hidden_states=None, attention_mask=None, position_ids=None, past_key_value=None,
                    output_attentions=False, use_cache=True):
            B, q_len, _ = hidden_states.shape
            q = self.q_proj(hidden_states).view(B, q_len, heads, D).transpose(1, 2)
            k = self.k_proj(hidden_states).view(B, q_len, heads, D).transpose(1, 2)
            kv = q_len + (0 if past_key_value.key_cache[self.layer_idx] is None else past_key_value.key_cache[self.layer_idx].shape[2])
            if new_style:
                cos, sin = self.rotary_emb(k, position_ids)
                cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)
            else:
                cos, sin = self.rotary_emb(k, seq_len=kv)
                cos, sin = cos[position_ids].unsqueeze(1), sin[position_ids].unsqueeze(1)
            rh = lambda x: torch.cat((-x[..., D // 2:], x[..., :D // 2]), dim=-1)
            q, k = (q * cos) + (rh(q) * sin), (k * cos) + (rh(k) * sin)
            k = past_key_value.update(k, self.layer_idx)
            w = None
            if output_attentions:                    # the eager path of 4.37.2
                w = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(D) + attention_mask
                w = torch.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
            return hidden_states, w, past_key_value

    torch.manual_seed(9)
    layers = torch.nn.ModuleList([torch.nn.Module() for _ in range(2)])
    for i, l in enumerate(layers):
        l.self_attn = Attn(i).to(dev).half()
    model = types.SimpleNamespace(model=types.SimpleNamespace(layers=layers), config=types.SimpleNamespace(output_attentions=False))
    B, prompt, pads = 2, 620, [0, 9]
    starts = [5 + p for p in pads]
    ends = [s0 + 576 for s0 in starts]
    hidden = torch.randn(B, prompt + 2, hid, device=dev).half() * 0.5

    def run(register):
        hl = ae.BatchMaskHookLogger(model, dev, layer_index=1)
        register(hl)
        hl.set_batch_image_token_ranges(starts, ends)
        cache = Cache(2)
        done = 0
        for q_len in (prompt, 1, 1):
            kv = done + q_len
            pos = torch.stack([torch.arange(done, kv, device=dev) - p for p in pads]).clamp_min(0)
            mask = torch.zeros(B, 1, q_len, kv, device=dev, dtype=torch.float16)
            causal = torch.arange(kv, device=dev)[None, :] > (done + torch.arange(q_len, device=dev))[:, None]
            mask.masked_fill_(causal[None, None], torch.finfo(torch.float16).min)
            for b, p in enumerate(pads):
                mask[b, :, :, :p] = torch.finfo(torch.float16).min
            layers[1].self_attn(hidden_states=hidden[:, done:kv], attention_mask=mask, position_ids=pos,
                                past_key_value=cache, use_cache=True)
            done = kv
        maps = torch.stack(hl.finalize_batch())
        hl.remove_hook_and_unpatch()
        return maps, len(hl.step_attentions)

    ref, n_ref = run(lambda hl: hl.register_hook_and_patch())            # eager weights -> reference-style hook
    got, n_got = run(lambda hl: hl.register_probe_legacy())              # q_proj + rotary on one token + key cache
    assert n_ref == n_got == 3
    # the probe forms exact fp16 dot products where the eager matmul accumulates in its own order: same tolerance as
    # test_probe_equals_hooked_eager_on_hf_llama (layer 0)
    np.testing.assert_allclose(N(got.float()), N(ref.float()), rtol=4e-3, atol=2e-6)
    assert abs(float(got[0].float().sum()) - 1.0) < 2e-3


def test_bench_gpus2_self_launch_on_one_gpu(dev):
    """`python bench.py --gpus 2` brings up two ranks by itself (parent spawns before touching the GPU); on this one-GPU
    box both ranks share device 0 over gloo.  The JSON line must say n_gpus = 2 and carry both ranks' rates."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--device", "0",
                        "--workload", "336", "--steps", "5", "--warmup", "2"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["scaling"] == "weak"
    assert len(d["per_rank_images_per_s"]) == 2 and d["weights_broadcast"]["bytes"] == 2755074 * 4
    assert d["value"] > 0 and d["roofline"]["frac"] > 0
    # the 336x336 workloads time the HIP-graph path (no host call per kernel); it must equal the serial result
    assert d["bit_identical_to_serial"] is True and d["rccl_ranks_seen"] == [0, 1] and "also_eager" in d
    # the curve apart from GPU-to-GPU spread: job rate / (N x mean rank rate) -- 1.0 when the ranks are equally fast, below it
    # by exactly the spread (value divides by the slowest rank's time)
    assert len(d["per_rank_roofline_frac"]) == 2 and d["dist_backend"] == "gloo"
    rates = d["per_rank_images_per_s"]
    assert 0.5 < d["scaling_efficiency_vs_rank_mean"] <= 1.0 + 1e-6
    assert abs(d["scaling_efficiency_vs_rank_mean"] - d["value"] / (2 * sum(rates) / 2)) < 2e-3


def test_bench_gpus2_default_workload_runs_the_dist_legs_on_every_rank(dev):
    """world_size 2 through the REAL N > 1 path with GPU work: `python bench.py --gpus 2` on the default (1024 x 1024) workload,
    both ranks sharing this box's one GPU over gloo (`--device 0 --dist-backend gloo`: the explicit smoke-test override of the
    RCCL requirement).  The line must carry, from BOTH ranks, BASELINE configs[3]'s per-rank batch and the ragged main_batched
    leg -- barrier-bracketed, max over ranks, per-rank rates, bit-identity ANDed over the ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--device", "0",
                        "--steps", "6", "--warmup", "2", "--legs", "336,main_batched_ragged"], env=env, capture_output=True, text=True,
                       timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 512 and d["rccl_ranks_seen"] == [0, 1] and d["dist_backend"] == "gloo"
    assert d["bit_identical_to_serial"] is True and len(d["per_rank_images_per_s"]) == 2
    for key, per_rank in (("also_336x256", 256), ("also_main_batched_ragged", 32)):
        leg = d[key]
        assert leg["n_gpus"] == 2 and len(leg["per_rank_images_per_s"]) == 2 and leg["bit_identical_to_serial"] is True, key
        rates = leg["per_rank_images_per_s"]
        assert all(v > 0 for v in rates) and 0.3 < leg["scaling_efficiency_vs_rank_mean"] <= 1.0 + 1e-6, key
        assert abs(leg["scaling_efficiency_vs_rank_mean"] - leg["value"] / (2 * sum(rates) / 2)) < 2e-3, key
    assert d["also_336x256"]["global_batch"] == 512
    assert [c["B"] for c in d["also_main_batched_ragged"]["cases"]] == [32, 256]
    assert "also_main_batched" not in d and "also_u8" not in d          # one-GPU legs stay off an N > 1 line


_RCCL_ONE_RANK = r"""
import json, sys, torch
sys.path.insert(0, sys.argv[1])
from attwarp_amd import dist as D
from attwarp_amd.model import MarginalNet
import torch.distributed as dist
rank, world, local = D.init(force_group=True)                  # WORLD_SIZE unset: a ONE-rank nccl (= RCCL) group on cuda:0
assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == "nccl"
net = MarginalNet(1024, 4096, 256).cuda()
before = [p.detach().clone() for p in net.parameters()]
nbytes = D.broadcast_module_weights(net, src=0)                # device branch: flat.to(dev) + dist.broadcast over RCCL
same = all(torch.equal(a, b) for a, b in zip(before, net.parameters()))
g = D.all_gather_counters({"images": 256.0, "rank": float(rank)})   # device-side all_gather
m = D.max_over_ranks(3.25)                                     # device-side all_reduce(MAX)
D.barrier()                                                    # barrier(device_ids=[...])
torch.cuda.synchronize()
print(json.dumps({"bytes": nbytes, "same": same, "gather": g, "max": m, "backend": dist.get_backend()}))
D.shutdown()
assert not dist.is_initialized()
"""


def test_rccl_one_rank_group_runs_every_collective(dev):
    """the RCCL branches of dist.py (weight broadcast, all_gather, all_reduce, barrier) executed on the one GPU
    this box has, as a one-rank communicator in its own process (the process group is process-global state)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK, root], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d == {"bytes": 11020296, "same": True, "gather": {"images": [256.0], "rank": [0.0]}, "max": 3.25, "backend": "nccl"}


def test_bench_force_dist_goes_through_rccl_on_one_gpu(dev):
    """`bench.py --gpus 1 --force-dist`: the code of a rank of N (RCCL init, weight broadcast, gathered counters, max over
    ranks, barriers around the timed region) on the one GPU, with the fields the N > 1 line carries."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--workload", "336x256",
                        "--steps", "8", "--warmup", "2", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["dist_backend"] == "nccl" and d["rccl_ranks_seen"] == [0]
    assert d["weights_broadcast"]["bytes"] == 11020296
    assert len(d["per_rank_images_per_s"]) == 1 and len(d["per_rank_roofline_frac"]) == 1
    assert abs(d["scaling_efficiency_vs_rank_mean"] - 1.0) < 1e-3 and d["bit_identical_to_serial"] is True


def test_bench_force_dist_default_workload_carries_the_dist_legs(dev):
    """The command the driver runs for the scaling curve, on the one GPU of this box: `bench.py --gpus 1 --force-dist` with the
    DEFAULT (1024 x 1024) workload.  Behind the main line every rank of the group -- here the one rank of a one-rank RCCL group --
    runs BASELINE configs[3]'s per-rank batch (also_336x256) and the ragged main_batched leg with the per-rank fields of the
    main line, n_gpus == 1 semantics: one rate, efficiency 1, and the bit-identity checks of both legs."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "8", "--warmup", "2",
                        "--no-cpu-baseline", "--legs", "336,main_batched_ragged"], env=env, capture_output=True,
                       text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["dist_backend"] == "nccl" and d["rccl_ranks_seen"] == [0] and "driver" in d["scaling_curve"]
    assert d["config"]["image_size"] == 1024 and d["roofline"]["frac"] > 0.3
    for key, per_rank in (("also_336x256", 256), ("also_main_batched_ragged", 32)):
        leg = d[key]
        assert leg["n_gpus"] == 1 and len(leg["per_rank_images_per_s"]) == 1 and leg["bit_identical_to_serial"] is True, key
        assert abs(leg["scaling_efficiency_vs_rank_mean"] - 1.0) < 2e-3 and leg["value"] > 0, key
        assert abs(leg["value"] - leg["per_rank_images_per_s"][0]) <= 2e-3 * leg["value"], key
    assert d["also_336x256"]["global_batch"] == 256
    rag = d["also_main_batched_ragged"]
    assert [c["B"] for c in rag["cases"]] == [32, 256] and all(c["bit_identical_to_serial"] for c in rag["cases"])
    assert rag["step_over_copy"] is not None and rag["cases"][0]["calibration"]["ms"] > 0


def test_wide_rows_sorted_random_maps(dev):
    """Rows wider than one staged LDS row (4500 floats / bytes) with irregular but sorted maps: column tiles; same bits."""
    from attwarp_amd import checkpoint_utils as cu
    rng = np.random.default_rng(81)
    H, W, C = 12, 1500, 3                                  # 4500 floats per row
    for dt in (np.float32, np.uint8):
        img = rng.random((1, H, W, C), dtype=np.float32)
        if dt == np.uint8:
            img = (img * 255).astype(np.uint8)
        mx = np.sort(rng.random((1, 1600)).astype(np.float32) * W, axis=1)
        my = np.sort(rng.random((1, 20)).astype(np.float32) * H, axis=1)
        got = N(cu.remap_separable(T(img, dev), T(mx, dev), T(my, dev), channels_last=True))
        assert np.array_equal(got[0], O.remap_bilinear(img[0], mx[0], my[0]))


def test_empty_batch_behaves_like_reference(dev):
    """warp_from_cdf_torch on B=0: the reference ends in np.stack([]) -> ValueError."""
    from attwarp_amd import checkpoint_utils as cu
    with pytest.raises(ValueError, match="at least one"):
        cu.warp_from_cdf_torch(torch.zeros(0, 3, 8, 8, device=dev), torch.zeros(0, 8, device=dev),
                               torch.zeros(0, 8, device=dev))


@pytest.mark.parametrize("world", [2, 8])
def test_shard_equivalence(dev, world):
    """Multi-GPU correctness by construction: the batch is cut into contiguous per-rank blocks
    (dist.shard_range) and every rank runs the same pipeline on its block; concatenating the per-rank
    results must give the single-process result byte for byte (images are independent)."""
    from attwarp_amd import pipeline
    from attwarp_amd.dist import shard_range
    rng = np.random.default_rng(91)
    B, T_, heads, kv, S = 11, 3, 32, 640, 64
    rows = T(softmax_rows(rng, (T_, B, heads, kv), peak=(100, 140)), dev)
    starts = T((35 + np.arange(B) % 8).astype(np.int32), dev)
    img = T(rng.random((B, S, S, 3), dtype=np.float32), dev)
    full = pipeline.warp_from_attention_stack(img, rows, starts, channels_last=True)
    parts = []
    for r in range(world):
        lo, hi = shard_range(B, r, world)
        if hi > lo:
            parts.append(pipeline.warp_from_attention_stack(img[lo:hi].contiguous(), rows[:, lo:hi].contiguous(),
                                                            starts[lo:hi].contiguous(), channels_last=True))
    assert torch.equal(torch.cat(parts), full)


def test_config3_batch2048_shards_at_full_size(dev):
    """BASELINE configs[3] at its stated size on one GPU: 2048 float32 336x336 images with their attention stacks
    (T=20, 32 heads, kv=640), cut into the 8 contiguous per-rank blocks of 256 that `bench.py --gpus 8 --workload
    336x256` gives every rank.  Each block through the whole path equals the same rows of the unsharded run byte for
    byte, and a few images equal the CPU oracle."""
    from attwarp_amd import pipeline
    from attwarp_amd.dist import shard_range
    B, T_, S, world = 2048, 20, 336, 8
    g = torch.Generator(device=dev).manual_seed(2048)
    img = torch.rand((B, S, S, 3), device=dev, generator=g)
    rows = torch.empty((T_, B, 32, 640), device=dev)
    for t in range(T_):
        rows[t] = torch.softmax(torch.randn((B, 32, 640), device=dev, generator=g), dim=-1)
    starts = (35 + torch.arange(B, device=dev) % 8).to(torch.int32)
    full = pipeline.warp_from_attention_stack(img, rows, starts, channels_last=True)
    for r in range(world):
        lo, hi = shard_range(B, r, world)
        assert hi - lo == 256
        part = pipeline.warp_from_attention_stack(img[lo:hi], rows[:, lo:hi].contiguous(), starts[lo:hi].contiguous(),
                                                  channels_last=True)
        assert torch.equal(part, full[lo:hi]), r
    for b in (0, 1023, 2047):
        r_np = N(rows[:, b:b + 1])
        att = O.attn_reduce_stack(r_np, [int(starts[b])]).reshape(1, 1, 24, 24)
        px, py = O.gt_marginals(att)
        Fx = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(px, S), 0))
        Fy = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(py, S), 0))
        mx, my = O.maps_from_cdf(Fx, Fy)
        assert np.array_equal(N(full[b]), O.remap_bilinear(N(img[b]), mx[0], my[0])), b


@pytest.mark.parametrize("name", ["sq500", "sq336", "land", "small"])
def test_clip_preprocess_bit_exact(dev, golden, name):
    """"next" row 3: warped uint8 image -> CLIP tensor on the GPU == HF CLIPImageProcessor (PIL backend)."""
    from attwarp_amd import pipeline
    g = golden("clip_preprocess")
    img = clip_input(name)
    out = pipeline.clip_preprocess(T(np.stack([img, img[::-1].copy()]), dev), 336, torch.float32)
    assert out.shape == (2, 3, 336, 336)
    got = N(out)
    assert np.array_equal(got[0], O.clip_preprocess(img, 336))
    assert np.array_equal(got[1], O.clip_preprocess(img[::-1].copy(), 336))
    d = clip_digest(got[0])
    assert np.array_equal(d["sub"], g[f"{name}_sub"])
    assert int(d["sum_bits"]) == int(g[f"{name}_sum_bits"]) and int(d["wsum_bits"]) == int(g[f"{name}_wsum_bits"])
    half = pipeline.clip_preprocess(T(img[None], dev), 336)                # LLaVA feeds float16
    assert half.dtype == torch.float16
    assert np.array_equal(N(half[0]), got[0].astype(np.float16))


@pytest.mark.parametrize("shape", [
    (3, 500, 500, 3, 336),     # the main_batched output: 7 taps, rows of 1500 bytes (dword staging)
    (2, 333, 517, 3, 224),     # landscape with crop, source rows not a multiple of 4 bytes (byte staging)
    (2, 517, 333, 3, 224),     # portrait
    (2, 1100, 1100, 3, 336),   # 15 taps: the 16-coefficient variant
    (1, 2000, 1900, 3, 336),   # 25 taps: generic kernels
    (2, 400, 400, 1, 336),     # one channel
    (2, 400, 420, 4, 336),     # four channels
    (2, 300, 300, 3, 335),     # output rows of 1005 bytes: generic kernels
    (2, 200, 260, 3, 336),     # up-scaling
])
def test_clip_preprocess_staged_equals_generic(dev, shape):
    """The LDS-staged kernels against the one-thread-per-output form (itself pinned to the HF processor by the
    goldens) and, for RGB, against the oracle: bit-for-bit, float32 and float16."""
    from attwarp_amd import pipeline
    B, H, W, C, size = shape
    rng = np.random.default_rng(H * 31 + W)
    img = rng.integers(0, 256, (B, H, W, C), dtype=np.uint8)
    mean, std = (0.4, 0.5, 0.6, 0.7)[:C], (0.2, 0.3, 0.25, 0.35)[:C]
    x = T(img, dev)
    for dt in (torch.float32, torch.float16):
        a = pipeline.clip_preprocess(x, size, dt, mean, std)
        with _lib.debug_override(clip_variant=1):
            b = pipeline.clip_preprocess(x, size, dt, mean, std)
        assert a.shape == (B, C, size, size) and torch.equal(a, b), (shape, dt)
    if C == 3 and H * W <= 600 * 600:
        got = N(pipeline.clip_preprocess(x, size, torch.float32))
        assert np.array_equal(got[0], O.clip_preprocess(img[0], size))


@pytest.mark.parametrize("hw", [(300, 420), (421, 300), (336, 336)])
def test_clip_preprocess_pad_to_square(dev, hw):
    """LLaVA-1.5's image_aspect_ratio="pad": expand2square (mean-colour canvas, image centred) in front of the CLIP
    processor arithmetic; wide, tall (odd margin) and already-square images against the oracle, bit for bit."""
    from attwarp_amd import pipeline
    H, W = hw
    rng = np.random.default_rng(H + 3 * W)
    img = rng.integers(0, 256, (2, H, W, 3), dtype=np.uint8)
    x = T(img, dev)
    sq = pipeline.expand2square(x)
    for b in range(2):
        assert np.array_equal(N(sq)[b], O.expand2square(img[b]))
    assert (sq is x) == (H == W)
    got = N(pipeline.clip_preprocess(x, 336, torch.float32, pad_to_square=True))
    for b in range(2):
        assert np.array_equal(got[b], O.clip_preprocess(O.expand2square(img[b]), 336))
    bg = (7, 8, 9)
    assert np.array_equal(N(pipeline.expand2square(x, bg))[1], O.expand2square(img[1], bg))


def test_warp_to_clip_pipeline(dev, golden):
    """main_batched chain + CLIP epilogue with nothing leaving the GPU: masks -> 500x500 uint8 warp -> [B,3,336,336]."""
    from attwarp_amd import pipeline
    g = golden("mask_postproc")
    rng = np.random.default_rng(93)
    imgs = rng.integers(0, 256, (4, 336, 336, 3), dtype=np.uint8)
    warped = pipeline.warp_from_masks(T(imgs, dev), T(g["masks"], dev), (500, 500))
    clip = pipeline.clip_preprocess(warped, 336, torch.float32)
    w = N(warped)
    for b in range(4):
        assert np.array_equal(N(clip[b]), O.clip_preprocess(w[b], 336))


@pytest.mark.parametrize("name", ["same", "up"])
def test_marginalnet_tail_kernels(dev, golden, name):
    """"next" row 1: masked token mean and FiLM + axis means -- bit-exact vs the oracle, float32-sum-order close
    to the tensors hooked out of the reference model, and the whole forward (library GEMMs + these kernels +
    safe_softmax) against the reference's (px, py)."""
    from attwarp_amd import model
    g = golden("marginalnet_tail")
    tok, mask = g[f"{name}_ttok"], g[f"{name}_tmask"]
    t = N(model.masked_token_mean(T(tok, dev), T(mask, dev)))
    assert np.array_equal(t, O.masked_token_mean(tok, mask[..., 0]))
    np.testing.assert_allclose(t, g[f"{name}_tmean"], rtol=0, atol=2e-7)
    for dt in (torch.float16, torch.bfloat16):                      # the reference's .float() happens in the kernel
        td = T(tok, dev).to(dt)
        assert np.array_equal(N(model.masked_token_mean(td, T(mask, dev))),
                              O.masked_token_mean(N(td.float()), mask[..., 0]))
    v, gb = g[f"{name}_v"], g[f"{name}_gamma_beta"]
    vx, vy = model.film_axis_means(T(v, dev), T(gb, dev))
    ox, oy = O.film_axis_means(v, gb)
    assert np.array_equal(N(vx), ox) and np.array_equal(N(vy), oy)
    np.testing.assert_allclose(N(vx), g[f"{name}_vx"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(N(vy), g[f"{name}_vy"], rtol=0, atol=2e-7)
    H, W = (int(x) for x in g[f"{name}_HW"])
    net = model.MarginalNet(20, 40, 16).eval()
    model.load_reference_checkpoint(net, {k.split("|", 1)[1]: torch.from_numpy(g[k]) for k in g.files
                                          if k.startswith(f"{name}_sd|")})
    net = net.to(dev)
    with torch.no_grad():
        px, py = net(T(g[f"{name}_fmap"], dev), H, W, T(tok, dev), T(mask, dev))
        lx, ly = net.forward_logits(T(g[f"{name}_fmap"], dev), H, W, T(tok, dev), T(mask, dev))
        fx, fy = net.forward_logits_fused(T(g[f"{name}_fmap"], dev), H, W, T(tok, dev), T(mask, dev))
    np.testing.assert_allclose(N(px), g[f"{name}_px"], rtol=2e-4, atol=1e-7)      # MIOpen / rocBLAS vs CPU GEMMs
    np.testing.assert_allclose(N(py), g[f"{name}_py"], rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(N(fx), N(lx), rtol=1e-4, atol=1e-5)                # fused tail vs stock-op tail
    np.testing.assert_allclose(N(fy), N(ly), rtol=1e-4, atol=1e-5)


def test_marginalnet_tail_shapes(dev):
    from attwarp_amd import model, _lib
    rng = np.random.default_rng(9)
    for (B, Ch, H, W) in [(3, 5, 24, 24), (2, 7, 17, 33), (1, 3, 63, 64), (5, 2, 1, 9)]:
        v = rng.standard_normal((B, Ch, H, W)).astype(np.float32)
        gb = rng.standard_normal((B, 2 * Ch)).astype(np.float32)
        vx, vy = model.film_axis_means(T(v, dev), T(gb, dev))
        ox, oy = O.film_axis_means(v, gb)
        assert np.array_equal(N(vx), ox) and np.array_equal(N(vy), oy), (B, Ch, H, W)
    with pytest.raises(_lib.AttWarpError, match="4096"):
        model.film_axis_means(torch.zeros(1, 1, 64, 64, device=dev), torch.zeros(1, 2, device=dev))
    for (B, Lt, D) in [(2, 1, 5), (3, 37, 300), (1, 8, 4096)]:
        tok = rng.standard_normal((B, Lt, D)).astype(np.float32)
        mask = (rng.random((B, Lt)) > 0.4).astype(np.float32)
        mask[0] = 0                                                       # fully masked sample -> zeros
        t = N(model.masked_token_mean(T(tok, dev), T(mask, dev)))
        assert np.array_equal(t, O.masked_token_mean(tok, mask))
        assert not t[0].any()


def test_config5_data_flow_with_random_clip_tower(dev):
    """BASELINE configs[4] as far as it can be built here: uint8 images -> CLIP tensor -> ViT-L/14-336 vision tower (the
    architecture of LLaVA-1.5's, SEEDED RANDOM weights: the checkpoint is absent) -> hidden_states[-2][:, 1:] ->
    [B,1024,24,24] -> MarginalNet(1024, 4096, 256) -> px, py -> warp -> CLIP tensor -> tower.  The product's CLIP tensors
    equal the HF processor's bit for bit (float32; float16 = its rounding), so the tower sees identical inputs and returns
    identical features; the hand-written legs inside the chain equal their stand-alone calls.  TextVQA accuracy parity is
    unobtainable (no weights, no dataset)."""
    from transformers.models.clip.image_processing_pil_clip import CLIPImageProcessorPil
    from attwarp_amd import pipeline
    from attwarp_amd.model import MarginalNet
    tower = pipeline.random_clip_vision_tower(dev, torch.float16, seed=0)
    torch.manual_seed(5)
    net = MarginalNet(1024, 4096, 256).to(dev).eval()
    imgs = np.stack([clip_input("sq500"), clip_input("sq500")[::-1, ::-1].copy()])
    x = T(imgs, dev)
    g = torch.Generator(device="cpu").manual_seed(6)
    txt = torch.randn(2, 12, 4096, generator=g).to(dev)
    mask = torch.ones(2, 12, 1, device=dev); mask[1, 7:] = 0          # [B,Lt,1] as the reference passes it (MN/model.py:77)
    r = pipeline.config5_chain(tower, net, x, txt, mask, out_size=(500, 500))
    assert r["token_map"].shape == (2, 1024, 24, 24) and r["features_warped"].shape == (2, 1024, 24, 24)
    assert r["px"].shape == (2, 24) and abs(float(r["px"][0].sum()) - 1.0) < 1e-5
    assert r["warped"].shape == (2, 500, 500, 3) and r["warped"].dtype == torch.uint8
    assert bool(torch.isfinite(r["features_warped"].float()).all())
    # the HF processor (PIL backend, as LLaVA-1.5 configures it) on the same uint8 images: identical tensors
    proc = CLIPImageProcessorPil(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336})
    hf_in = proc.preprocess([im for im in imgs], return_tensors="np")["pixel_values"]
    assert np.array_equal(N(pipeline.clip_preprocess(x, 336, torch.float32)), hf_in)
    assert torch.equal(r["pixel_values_in"], T(hf_in, dev).half())
    hf_w = proc.preprocess([w for w in N(r["warped"])], return_tensors="np")["pixel_values"]
    assert torch.equal(r["pixel_values_warped"], T(hf_w, dev).half())
    # identical inputs (asserted above: that is the product's part) -> the same tower features; the tower is library GEMM /
    # attention kernels, bit-identical from run to run on every lease so far -- a different algorithm choice between two
    # calls would still have to agree to float16 accuracy
    def same(a, b):
        return bool(torch.equal(a, b)) or bool(torch.allclose(a.float(), b.float(), rtol=2e-3, atol=2e-3))
    assert same(pipeline.vision_tower_features(tower, T(hf_in, dev).half()), r["token_map"])
    assert same(pipeline.vision_tower_features(tower, T(hf_w, dev).half()), r["features_warped"])
    # the legs inside the chain are the stand-alone calls
    with torch.no_grad():
        px, py = net(r["token_map"], 24, 24, txt, mask)
    # (MarginalNet's convolutions are library kernels: the algorithm MIOpen picks may differ between two calls)
    assert torch.allclose(px, r["px"], rtol=1e-4, atol=1e-7) and torch.allclose(py, r["py"], rtol=1e-4, atol=1e-7)
    assert torch.equal(pipeline.warp_from_pdf(x, r["px"], r["py"], (500, 500), channels_last=True), r["warped"])


def test_marginalnet_to_warp_chain(dev, golden):
    """Config-5 chain at toy size: seeded reference weights -> MarginalNet on the GPU -> maps -> warp.
    The network runs on library convolutions (tolerance on px, py); given ITS px, py the rest is bit-exact."""
    from attwarp_amd import model, pipeline
    g = golden("marginalnet")
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd|")}
    net = model.MarginalNet(32, 48, hidden=16).eval()
    net.load_state_dict(sd)
    net = net.to(dev)
    rng = np.random.default_rng(95)
    img = rng.random((2, 3, 96, 128), dtype=np.float32)
    out, px, py = pipeline.warp_from_marginalnet(net, T(g["fmap"], dev), T(g["ttok"], dev), T(g["tmask"], dev), T(img, dev))
    np.testing.assert_allclose(N(px), g["px"], rtol=1e-4, atol=1e-6)
    Fx = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(N(px), 128), 0))
    Fy = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(N(py), 96), 0))
    assert np.array_equal(N(out), O.warp_from_cdf(img, Fx, Fy))


def test_remap_fuzz_shapes(dev):
    """Seeded random shapes / channel counts / layouts / dtypes / map kinds through every dispatch path
    (float rows kernel, uint8 rows kernel, gather fallback for unaligned shapes): bit-exact vs the oracle."""
    from attwarp_amd import checkpoint_utils as cu
    rng = np.random.default_rng(2026)
    for case in range(60):
        H, W = int(rng.integers(24, 160)), int(rng.integers(24, 200))
        Ho, Wo = int(rng.integers(1, 180)), int(rng.integers(1, 220))
        C = int(rng.integers(1, 5))
        if case % 3 == 0:                     # aligned shapes so the fast kernels are exercised, not only the fallback
            W, Wo = (W // 4) * 4 + 4, (Wo // 4) * 4 + 4
        B = int(rng.integers(1, 4))
        kind = ("cdf", "wild", "identity")[case % 3]
        mx, my = make_maps(rng, B, H, W, Ho, Wo, kind)
        dt = (np.float32, np.uint8)[case % 2]
        img = rng.random((B, H, W, C), dtype=np.float32)
        if dt == np.uint8:
            img = (img * 255).astype(np.uint8)
        mode = ("exact", "cv2")[(case // 2) % 2]
        ref = np.stack([O.remap_bilinear(img[b], mx[b], my[b], mode) for b in range(B)])
        hwc = N(cu.remap_separable(T(img, dev), T(mx, dev), T(my, dev), mode=mode, channels_last=True))
        chw = N(cu.remap_separable(T(img.transpose(0, 3, 1, 2), dev), T(mx, dev), T(my, dev), mode=mode))
        tag = (case, H, W, Ho, Wo, C, kind, dt.__name__, mode)
        assert np.array_equal(hwc, ref), tag
        assert np.array_equal(chw.transpose(0, 2, 3, 1), ref), tag


@pytest.mark.parametrize("S,B,layout,mode,adt", [(96, 5, "hwc", "cv2", torch.float32), (336, 3, "hwc", "cv2", torch.float32),
                                                  (336, 2, "chw", "exact", torch.float32), (1024, 2, "hwc", "cv2", torch.float32),
                                                  (1024, 1, "hwc", "exact", torch.float32), (1024, 1, "chw", "cv2", torch.float32),
                                                  (512, 2, "hwc", "cv2", torch.float32), (200, 3, "chw", "cv2", torch.float32),
                                                  (336, 3, "hwc", "cv2", torch.float16), (336, 2, "chw", "cv2", torch.bfloat16),
                                                  (1024, 1, "hwc", "cv2", torch.float16), (96, 4, "hwc", "exact", torch.bfloat16)])
def test_fused_step_equals_three_launches(dev, S, B, layout, mode, adt):
    """attwarp_warp_step_fused (resample of batch k + maps of batch k+1 + attention reduce of batch k+2 as block ranges
    of ONE launch) against the three separate entry points, bit for bit, on every staged resample family; and its
    argument checks."""
    from attwarp_amd import pipeline, checkpoint_utils as cu, attention_extraction as ae
    g = torch.Generator(device=dev).manual_seed(S + B)
    T = 5
    cl = layout == "hwc"
    imgs = [torch.rand((B, S, S, 3) if cl else (B, 3, S, S), device=dev, generator=g) for _ in range(3)]
    rws = [torch.softmax(torch.randn((T, B, 32, 640), device=dev, generator=g) * (1 + 2 * k), dim=-1).to(adt) for k in range(3)]
    starts = (35 + torch.arange(B, device=dev) % 8).to(torch.int32)
    ow = pipeline.OverlappedWarp(imgs, rws, starts, channels_last=cl, mode=mode, pattern="fused")
    assert ow.steps[0].dtype == adt
    assert ow.pattern == "fused"
    # one fused step by hand: R(0) with the maps of batch 0, M from the steps of batch 1, A on the rows of batch 2
    steps0 = pipeline.attention_step_maps(rws[0], starts)
    steps1 = pipeline.attention_step_maps(rws[1], starts)
    steps2 = pipeline.attention_step_maps(rws[2], starts)
    m0 = pipeline.axis_maps_from_attention_steps(steps0, (S, S))
    m1 = pipeline.axis_maps_from_attention_steps(steps1, (S, S))
    ref0 = cu.remap_separable(imgs[0], *m0, mode=mode, channels_last=cl)
    ow.steps[0].copy_(steps1); ow.maps[0][0].copy_(m0[0]); ow.maps[0][1].copy_(m0[1])
    ow.steps[1].fill_(-1); ow.maps[1][0].fill_(-1); ow.maps[1][1].fill_(-1); ow.outs[0].fill_(-1)
    ow._fused_step(0, 0)
    assert torch.equal(ow.outs[0], ref0)
    assert torch.equal(ow.maps[1][0], m1[0]) and torch.equal(ow.maps[1][1], m1[1])
    assert torch.equal(ow.steps[1], steps2)
    # a whole stream through the graph path
    refs = [pipeline.warp_from_attention_stack(imgs[k], rws[k], starts, channels_last=cl, mode=mode) for k in range(3)]
    for K in (7, 4):
        ow.reset(); ow.prime(); ow.prime2(); ow.run(K - 2); ow.tail()
        for r in range(3):
            assert torch.equal(ow.outs[r], refs[r]), (K, r)


@pytest.mark.parametrize("dt", [np.uint8, np.float32])
def test_resize_linear_cv2_vs_oracle(dev, dt):
    """cv2.resize(INTER_LINEAR) (AGW/new_method.py:369): attwarp_resize_linear == the oracle's restatement of OpenCV's
    published arithmetic, bit for bit: up- and down-scaling, the exact 2 x 2 decimation (INTER_AREA), non-integer
    factors, thin images, 1 / 3 / 4 channels and the 2-D form; the exact mode stays available."""
    from attwarp_amd import new_method as nm
    rng = np.random.default_rng(77)
    for (H, W, C, th, tw) in [(37, 53, 3, 74, 106), (64, 48, 3, 32, 24), (37, 53, 1, 20, 30), (50, 70, 4, 18, 26),
                              (9, 200, 3, 90, 7), (336, 336, 3, 500, 500), (24, 24, 0, 336, 336), (31, 17, 3, 31, 18)]:
        shape = (H, W) if C == 0 else (H, W, C)
        img = rng.integers(0, 256, shape, dtype=np.uint8) if dt == np.uint8 else rng.random(shape, dtype=np.float32)
        att = np.zeros((th, tw), np.float32)
        got = nm.resize_image_to_match_attmap(img, att)
        assert got.dtype == dt and got.shape[:2] == (th, tw)
        assert np.array_equal(got, O.resize_linear_cv2(img, (tw, th))), (H, W, C, th, tw)
        ex = nm.resize_image_to_match_attmap(img, att, mode="exact")
        assert np.abs(ex.astype(np.float64) - got.astype(np.float64)).max() <= (1.0 if dt == np.uint8 else 1e-5)
    assert np.array_equal(nm.resize_image_to_match_attmap(img, np.zeros(img.shape[:2])), img)
    with pytest.raises(TypeError):
        nm.resize_image_to_match_attmap(img.astype(np.float64), att)


@pytest.mark.parametrize("mode", ["cv2", "exact"])
def test_remap_float64_pass_through(dev, mode):
    """warp_from_cdf_torch keeps the image dtype into cv2.remap (MN/checkpoint_utils.py:152,203): a float64 image is
    resampled in double (OpenCV's CV_64F arithmetic: float32 table weights, products accumulated in double), not
    narrowed to float32 -- kernel == oracle bit for bit, both layouts; and it differs from the float32 result."""
    from attwarp_amd import checkpoint_utils as cu
    rng = np.random.default_rng(5)
    B, C, H, W, Ho, Wo = 2, 3, 33, 47, 40, 52
    img = rng.random((B, C, H, W))
    Fx = np.cumsum(rng.random((B, W)).astype(np.float32) + 0.05, 1); Fx = (Fx / Fx[:, -1:]).astype(np.float32)
    Fy = np.cumsum(rng.random((B, H)).astype(np.float32) + 0.05, 1); Fy = (Fy / Fy[:, -1:]).astype(np.float32)
    got = cu.warp_from_cdf_torch(T(img, dev), T(Fx, dev), T(Fy, dev), (Ho, Wo), mode=mode)
    assert got.dtype == torch.float64
    ref = O.warp_from_cdf(img, Fx, Fy, (Ho, Wo), mode)
    assert ref.dtype == np.float64 and np.array_equal(N(got), ref)
    narrow = N(cu.warp_from_cdf_torch(T(img.astype(np.float32), dev), T(Fx, dev), T(Fy, dev), (Ho, Wo), mode=mode))
    assert 0 < np.abs(narrow - ref).max() < 1e-6
    mx, my = cu.axis_maps_from_cdf(T(Fx, dev), T(Fy, dev), (Ho, Wo))
    hwc = cu.remap_separable(T(np.ascontiguousarray(img.transpose(0, 2, 3, 1)), dev), mx, my, mode=mode, channels_last=True)
    assert np.array_equal(N(hwc).transpose(0, 3, 1, 2), ref)


@pytest.mark.parametrize("adt", [torch.float32, torch.float16, torch.bfloat16])
def test_attn_reduce_and_maps_equals_two_launches(dev, adt):
    """attwarp_attn_reduce_and_maps (reduce of batch k+2 + maps of batch k+1 as one launch) == attn_reduce_step +
    axis_maps_from_attention_steps, bit for bit, in every attention dtype; pattern auto picks it for large images."""
    from attwarp_amd import pipeline
    g = torch.Generator(device=dev).manual_seed(3)
    B, T, S = 3, 5, 128
    imgs = [torch.rand((B, S, S, 3), device=dev, generator=g) for _ in range(3)]
    rws = [torch.softmax(torch.randn((T, B, 32, 640), device=dev, generator=g) * (1 + k), dim=-1).to(adt) for k in range(3)]
    starts = (35 + torch.arange(B, device=dev) % 8).to(torch.int32)
    ow = pipeline.OverlappedWarp(imgs, rws, starts, channels_last=True, pattern="am")
    steps1 = pipeline.attention_step_maps(rws[1], starts)
    steps2 = pipeline.attention_step_maps(rws[2], starts)
    m1 = pipeline.axis_maps_from_attention_steps(steps1, (S, S))
    ow.steps[0].copy_(steps1); ow.steps[1].fill_(-1); ow.maps[1][0].fill_(-1); ow.maps[1][1].fill_(-1)
    ow._am_launch(0, 0)
    assert torch.equal(ow.steps[1], steps2) and torch.equal(ow.maps[1][0], m1[0]) and torch.equal(ow.maps[1][1], m1[1])
    refs = [pipeline.warp_from_attention_stack(imgs[k], rws[k], starts, channels_last=True) for k in range(3)]
    ow.reset(); ow.prime(); ow.prime2(); ow.run(9); ow.tail()
    for r in range(3):
        assert torch.equal(ow.outs[r], refs[r]), r
    big = torch.rand((1, 1024, 1024, 3), device=dev)
    assert pipeline.OverlappedWarp(big, rws[0][:, :1].contiguous(), starts[:1], channels_last=True).pattern == "am"
    assert pipeline.OverlappedWarp(imgs[0], rws[0], starts, channels_last=True).pattern == "fused"
    # uint8 images are not eligible for the one-launch step: auto takes the two-launch form, same results
    if adt == torch.float32:
        imgs8 = [(im * 255).to(torch.uint8) for im in imgs]
        ow8 = pipeline.OverlappedWarp(imgs8, rws, starts, channels_last=True)
        assert ow8.pattern == "am"
        ow8.prime(); ow8.prime2(); ow8.run(7); ow8.tail()
        for r in range(3):
            assert torch.equal(ow8.outs[r], pipeline.warp_from_attention_stack(imgs8[r], rws[r], starts, channels_last=True)), r


def test_fused_step_argument_checks(dev):
    from attwarp_amd._lib import call, ptr, AttWarpError
    z = torch.zeros(64, device=dev)
    with pytest.raises(AttWarpError, match="null image"):
        call("attwarp_warp_step_fused", None, ptr(z), 0, 1, 3, 4, 4, 4, 4, ptr(z), ptr(z), 1, 0, None, 0, 0, None, None, None,
             None, None, 0, 0, 0, None, 0, 0, None, None)
    with pytest.raises(AttWarpError, match="generic resample"):      # 5 floats per row: not a staged shape
        call("attwarp_warp_step_fused", ptr(z), ptr(z), 0, 1, 1, 2, 5, 2, 5, ptr(z), ptr(z), 1, 0, None, 0, 0, None, None, None,
             None, None, 0, 0, 0, None, 0, 0, None, None)
    with pytest.raises(AttWarpError, match="multiple of 4"):
        call("attwarp_warp_step_fused", ptr(z), ptr(z), 0, 1, 1, 2, 8, 2, 8, ptr(z), ptr(z), 1, 0, None, 0, 0, None, None, None,
             None, ptr(z), 1, 2, 30, ptr(z.int()), 1, 6, ptr(z), None)
    with pytest.raises(AttWarpError, match="attn_dtype"):
        call("attwarp_warp_step_fused", ptr(z), ptr(z), 0, 1, 1, 2, 8, 2, 8, ptr(z), ptr(z), 1, 3, None, 0, 0, None, None, None,
             None, ptr(z), 1, 2, 30, ptr(z.int()), 1, 8, ptr(z), None)


@pytest.mark.parametrize("pattern", ["fused", "am", "dag", "join"])
def test_overlapped_warp_equals_serial(dev, pattern):
    """pipeline.OverlappedWarp (resample of batch k || maps of batch k+1 || attention reduce of batch k+2 as one fused
    launch per step, or as branches of one HIP graph): every step is bit-identical to warp_from_attention_stack on the
    batch it belongs to, with the inputs refilled between steps per the documented protocol (attention runs two
    batches ahead)."""
    from attwarp_amd import pipeline
    import functools
    OW = functools.partial(pipeline.OverlappedWarp, pattern=pattern)
    g = torch.Generator(device=dev).manual_seed(21)
    B, T, S, NB = 6, 4, 96, 5
    imgs = [torch.rand((B, S, S, 3), device=dev, generator=g) for _ in range(NB)]
    rws = [torch.softmax(torch.randn((T, B, 32, 640), device=dev, generator=g) * (1 + k), dim=-1) for k in range(NB)]
    starts = (35 + torch.arange(B, device=dev) % 8).to(torch.int32)
    refs = [pipeline.warp_from_attention_stack(imgs[k], rws[k], starts, channels_last=True) for k in range(NB)]
    img, rows = imgs[0].clone(), rws[0].clone()
    ow = OW(img, rows, starts, channels_last=True)
    rows.copy_(rws[0]); ow.prime()
    rows.copy_(rws[1]); ow.prime2()
    for k in range(NB):
        img.copy_(imgs[k])
        if k + 2 < NB:
            rows.copy_(rws[k + 2])
        out = ow.step()
        assert torch.equal(out, refs[k]), k
    # steady state on static buffers (what bench.py times): unrolled graphs == serial
    ow2 = OW(imgs[1], rws[1], starts, channels_last=True)
    ow2.prime(); ow2.prime2()
    assert torch.equal(ow2.run(19), refs[1])
    # a RING of 3 buffers refilled by a producer that stays ahead: batch k lives in slot k % 3
    n, NS = 3, 11
    seq = [k % NB for k in range(NS)]
    ring_i = [torch.empty_like(imgs[0]) for _ in range(n)]
    ring_r = [torch.empty_like(rws[0]) for _ in range(n)]
    ow3 = OW(ring_i, ring_r, starts, channels_last=True)
    ring_r[0].copy_(rws[seq[0]]); ow3.prime()
    ring_r[1].copy_(rws[seq[1]]); ow3.prime2()
    for k in range(NS):
        ring_i[k % n].copy_(imgs[seq[k]])
        if k + 2 < NS:
            ring_r[(k + 2) % n].copy_(rws[seq[k + 2]])
        out = ow3.step()
        assert out is ow3.outs[k % n] and torch.equal(out, refs[seq[k]]), k
    # static ring (what bench.py times): prime, prime2, unrolled graphs over the ring, tail == serial per slot
    ring_i = [imgs[k].clone() for k in range(4)]
    ring_r = [rws[k].clone() for k in range(4)]
    ow4 = OW(ring_i, ring_r, starts, channels_last=True)
    for K in (13, 13, 6):
        ow4.reset(); ow4.prime(); ow4.prime2(); ow4.run(K - 2); ow4.tail()
        assert ow4.k == K
        for r in range(4):
            assert torch.equal(ow4.outs[r], refs[r]), (K, r)
    # prompts of another length from batch 2 on: set_starts before the step that reduces that batch
    starts_b = (starts + 5).contiguous()
    st = starts.clone()
    ow5 = OW(img, rows, st, channels_last=True)
    rows.copy_(rws[0]); ow5.prime()
    rows.copy_(rws[1]); ow5.prime2()
    for k in range(4):
        img.copy_(imgs[k])
        if k + 2 < 4:
            ow5.set_starts(starts_b)
            rows.copy_(rws[k + 2])
        out = ow5.step()
        ref = refs[k] if k < 2 else pipeline.warp_from_attention_stack(imgs[k], rws[k], starts_b, channels_last=True)
        assert torch.equal(out, ref), k
    with pytest.raises(ValueError):
        ow5.set_starts(starts_b[:-1])
    with pytest.raises(ValueError):
        OW(imgs[0].permute(0, 2, 1, 3), rws[0], starts, channels_last=True)
    steps = pipeline.attention_step_maps(rws[0], starts)
    buf = torch.empty(B, 2 * S, device=dev)
    with pytest.raises(ValueError):     # a strided view would be overwritten as if it were dense
        pipeline.axis_maps_from_attention_steps(steps, (S, S), maps_out=(buf[:, :S], torch.empty(B, S, device=dev)))


@pytest.mark.parametrize("pattern", ["serial", "branches", "fused"])
@pytest.mark.parametrize("case", [(3, 96, 80, 5), (2, 336, 500, 6), (5, 64, 64, 3)])
def test_mask_chain_stream_equals_warp_from_masks(dev, pattern, case):
    """pipeline.MaskChainStream (the main_batched chain as a batch stream) gives, for every batch of the stream, the bytes
    of pipeline.warp_from_masks on that batch -- for every pattern, over a ring shorter and longer than the pipeline depth,
    including the first (primed) and the last (drained) batches."""
    from attwarp_amd import pipeline
    B, S, So, n = case
    g = torch.Generator(device=dev).manual_seed(S + n)
    nb = 11
    imgs = [torch.randint(0, 256, (B, S, S, 3), device=dev, dtype=torch.uint8, generator=g) for _ in range(nb)]
    msk = [torch.rand(B, 24, 24, device=dev, generator=g) for _ in range(nb)]
    refs = [pipeline.warp_from_masks(imgs[j], msk[j], (So, So + 4)) for j in range(nb)]
    ring_i = [torch.empty_like(imgs[0]) for _ in range(n)]
    ring_m = [torch.empty_like(msk[0]) for _ in range(n)]
    try:
        mc = pipeline.MaskChainStream(ring_i, ring_m, (So, So + 4), pattern=pattern)
    except _lib.AttWarpError:
        assert pattern == "fused"
        pytest.skip("one-launch chain step not eligible for this shape")
    d = mc.depth
    if n < d + 1:
        pytest.skip("ring shorter than the pipeline")
    for j in range(min(d, nb)):
        ring_m[j % n].copy_(msk[j])
    mc.prime()
    got = []
    for k in range(nb - d):
        ring_i[k % n].copy_(imgs[k])
        if k + d < nb:
            ring_m[(k + d) % n].copy_(msk[k + d])
        got.append(mc.step().clone())
    # the tail: images of the last d batches, then the remaining stages serially
    for k in range(nb - d, nb):
        ring_i[k % n].copy_(imgs[k])
    mc.drain()
    for k in range(nb - d, nb):
        got.append(mc.outs[k % n].clone())
    for j in range(nb):
        assert torch.equal(got[j], refs[j]), (pattern, j)


@pytest.mark.parametrize("S,B,layout,mode,adt,n", [(96, 5, "hwc", "cv2", torch.float32, 4), (336, 3, "hwc", "cv2", torch.float16, 6),
                                                  (64, 2, "chw", "exact", torch.bfloat16, 2), (128, 4, "hwc", "exact", torch.float32, 8)])
def test_paired_step_warp_equals_serial(dev, S, B, layout, mode, adt, n):
    """pipeline.PairedStepWarp (one launch = R(2p), R(2p+1) | M(2p+2), M(2p+3) | A(2p+4), A(2p+5)) over a stream of 12
    DIFFERENT batches copied through rings of n static buffers: every batch equals warp_from_attention_stack on it."""
    from attwarp_amd import pipeline
    g = torch.Generator(device=dev).manual_seed(S + n)
    nb = 12
    hwc = layout == "hwc"
    shape = (B, S, S, 3) if hwc else (B, 3, S, S)
    imgs = [torch.rand(shape, device=dev, generator=g) for _ in range(nb)]
    rows = [torch.softmax(torch.randn(3, B, 8, 600, device=dev, generator=g) * 2, -1).to(adt) for _ in range(nb)]
    starts = (3 + torch.arange(B, device=dev) % 5).to(torch.int32)
    refs = [pipeline.warp_from_attention_stack(imgs[j], rows[j], starts, channels_last=hwc, mode=mode) for j in range(nb)]
    ring_i = [torch.empty_like(imgs[0]) for _ in range(n)]
    ring_r = [torch.empty_like(rows[0]) for _ in range(n)]
    pw = pipeline.PairedStepWarp(ring_i, ring_r, starts, channels_last=hwc, mode=mode)
    if n < 6:
        # the attention runs four batches ahead of the images and a launch reads two slots of each ring: with fewer than
        # six slots the copies below would overwrite rows a launch still needs -- only the steady state on static data
        for i in range(n):
            ring_i[i].copy_(imgs[i % 2]); ring_r[i].copy_(rows[i % 2])
        pw.prime(); pw.run(8); pw.tail()
        torch.cuda.synchronize()
        for i in range(n):
            assert torch.equal(pw.outs[i], refs[i % 2]), i
        return
    for j in range(4):
        ring_r[j % n].copy_(rows[j])
    pw.prime()
    got = {}
    for k in range(0, nb - 4, 2):
        for s in (0, 1):
            ring_i[(k + s) % n].copy_(imgs[k + s])
            ring_r[(k + 4 + s) % n].copy_(rows[k + 4 + s])
        pw.step()
        for s in (0, 1):
            got[k + s] = pw.outs[(k + s) % n].clone()
    for k in range(nb - 4, nb):
        ring_i[k % n].copy_(imgs[k])
    pw.tail()
    for k in range(nb - 4, nb):
        got[k] = pw.outs[k % n].clone()
    for j in range(nb):
        assert torch.equal(got[j], refs[j]), j


@pytest.mark.parametrize("shape", [(50, 50, 33, 47), (64, 683, 500, 500), (90, 77, 61, 88)])
@pytest.mark.parametrize("n", [1, 2, 4, 5])
def test_mask_chain_stream_unaligned_shapes_take_the_ragged_step(dev, shape, n):
    """Equally sized images whose width the uniform one-launch step refuses (W % 4 != 0) get the one-launch step of the
    ragged kernel through the SAME class (pattern "ragged", depth 4): rings of every length -- one static buffer included --
    graphs, primed and drained ends; every batch equals warp_from_masks."""
    from attwarp_amd import pipeline
    H, W, Wo, Ho = shape
    g = torch.Generator(device=dev).manual_seed(H + W + n)
    B = 3
    imgs = [torch.randint(0, 256, (B, H, W, 3), device=dev, dtype=torch.uint8, generator=g) for _ in range(n)]
    msk = [torch.rand(B, 24, 24, device=dev, generator=g) for _ in range(n)]
    mc = pipeline.MaskChainStream(imgs, msk, (Ho, Wo))
    assert mc.pattern == "ragged" and mc.depth == 4
    mc.prime(); mc.run(7); mc.drain()
    for j in range(n):
        assert torch.equal(mc.outs[j], pipeline.warp_from_masks(imgs[j], msk[j], (Ho, Wo))), j
    with pytest.raises(_lib.AttWarpError):
        pipeline.MaskChainStream(imgs, msk, (Ho, Wo), pattern="fused")


TRANSFORM_CASES = [("identity", 1.0, 1.0, True), ("square", 1.0, 1.0, False), ("square", 1.0, 1.0, True), ("sqrt", 1.0, 1.0, False),
                   ("sqrt", 1.0, 1.0, True), ("exp", 0.01, 2.0, False), ("exp", 0.02, 1.5, True), ("log", 1.0, 1.0, False),
                   ("log", 1.0, 1.0, True), ("bogus", 1.0, 1.0, False)]


@pytest.mark.parametrize("tr,es,ed,inv", TRANSFORM_CASES)
def test_one_launch_chains_take_save_warped_images_transforms(dev, tr, es, ed, inv):
    """save_warped_image's own keyword arguments (new_method.py:405-411: transform, exp_scale, exp_divisor, apply_inverse; :134-191
    the five transforms, default "sqrt") on the ONE-LAUNCH steps -- the uniform step (attwarp_mask_chain_step), the ragged step
    on equally sized unaligned images and on a batch of differently sized images, the ragged stream -- equal the five serial
    launches of warp_from_masks with the same arguments BIT FOR BIT (sqrt / exp / log read the table
    attwarp_attention_transform_lut wrote with the stand-alone kernel's device functions), and the serial launches equal the
    oracle given the revised mask (exp / log: the device libm may move a map entry by one float32 ulp)."""
    from attwarp_amd import pipeline, attention_extraction as ae
    kw = dict(transform=tr, exp_scale=es, exp_divisor=ed, apply_inverse=inv)
    g = torch.Generator(device=dev).manual_seed(len(tr) + int(inv))
    n, B = 6, 3
    # (a) the uniform one-launch step
    imgs = [torch.randint(0, 256, (B, 96, 100, 3), device=dev, dtype=torch.uint8, generator=g) for _ in range(n)]
    msk = [torch.rand(B, 24, 24, device=dev, generator=g) ** 2 for _ in range(n)]
    mc = pipeline.MaskChainStream(imgs, msk, (80, 88), **kw)
    assert mc.pattern == "fused"
    mc.prime(); mc.run(2 * n - mc.depth); mc.drain()
    want = [pipeline.warp_from_masks(imgs[j], msk[j], (80, 88), **kw) for j in range(n)]
    for j in range(n):
        assert torch.equal(mc.outs[j], want[j]), j
    # (b) the ragged step on equally sized images the uniform step refuses (97 pixels: rows of 291 bytes)
    imgs_u = [torch.randint(0, 256, (B, 70, 97, 3), device=dev, dtype=torch.uint8, generator=g) for _ in range(n)]
    mu = pipeline.MaskChainStream(imgs_u, msk, (80, 88), **kw)
    assert mu.pattern == "ragged"
    mu.prime(); mu.run(2 * n - mu.depth); mu.drain()
    for j in range(n):
        assert torch.equal(mu.outs[j], pipeline.warp_from_masks(imgs_u[j], msk[j], (80, 88), **kw)), j
    # (c) a batch of differently sized images: five ragged launches, and the ragged stream
    sizes = [(70, 61), (52, 96), (96, 75), (130, 40)]
    rag_i = [torch.randint(0, 256, (h, w, 3), device=dev, dtype=torch.uint8, generator=g) for (h, w) in sizes]
    att = torch.rand(len(sizes), 24, 24, device=dev, generator=g) ** 2
    rag = pipeline.warp_from_masks_ragged(rag_i, att, (80, 88), **kw)
    for b in range(len(sizes)):
        assert torch.equal(rag[b], pipeline.warp_from_masks(rag_i[b][None], att[b:b + 1], (80, 88), **kw)[0]), b
    st = pipeline.RaggedMaskChainStream(out_size=(80, 88), **kw)
    outs = [st.push(rag_i, att) for _ in range(6)]
    outs = [o for o in outs if o is not None] + st.flush()
    assert len(outs) == 6 and all(torch.equal(o.out, rag) for o in outs)
    # (d) the oracle, given the GPU's revised mask (the x255 truncation is sensitive to its last ulp: DESIGN 4)
    rev = N(ae.revise_mask(att, 3, 10))
    for b, (h, w) in enumerate(sizes):
        mota = O.lanczos_resize_u8(O.mask_to_u8(rev[b]), w, h)
        ref = O.warp_image_by_attention(N(rag_i[b]), mota, 88, 80, tr if tr != "bogus" else "identity", es, ed, inv)
        got = N(rag[b])
        if tr in ("exp", "log"):
            assert (got != ref).mean() <= 2e-3 and np.abs(got.astype(int) - ref.astype(int)).max() <= 255, (b, tr)
        else:
            assert np.array_equal(got, ref), (b, tr)


def test_warp_from_masks_ragged_argument_checks_and_exact_mode(dev):
    """warp_from_masks_ragged validates up front: an empty batch and images of different channel counts raise
    ValueError with a clear message instead of an IndexError / a shape mismatch deep inside; `mode="exact"` -- which the ragged
    kernel (the integer cv2 resample) does not compute -- runs image by image through warp_from_masks and equals it."""
    from attwarp_amd import pipeline
    g = torch.Generator(device=dev).manual_seed(3)
    imgs = [torch.randint(0, 256, (h, w, 3), device=dev, dtype=torch.uint8, generator=g) for (h, w) in ((40, 61), (52, 48))]
    att = torch.rand(2, 24, 24, device=dev, generator=g)
    with pytest.raises(ValueError, match="empty"):
        pipeline.warp_from_masks_ragged([], att[:0], (32, 36))
    with pytest.raises(ValueError, match="channel"):
        pipeline.warp_from_masks_ragged([imgs[0], imgs[1][:, :, :1].contiguous()], att, (32, 36))
    with pytest.raises(ValueError, match="map per image"):
        pipeline.warp_from_masks_ragged(imgs, att[:1], (32, 36))
    with pytest.raises(ValueError, match="mode"):
        pipeline.warp_from_masks_ragged(imgs, att, (32, 36), mode="nearest")
    ex = pipeline.warp_from_masks_ragged(imgs, att, (32, 36), mode="exact", transform="sqrt")
    for b in range(2):
        assert torch.equal(ex[b], pipeline.warp_from_masks(imgs[b][None], att[b:b + 1], (32, 36), mode="exact", transform="sqrt")[0])
    assert not torch.equal(ex, pipeline.warp_from_masks_ragged(imgs, att, (32, 36), transform="sqrt"))      # (cv2 arithmetic differs)


def test_ragged_chain_p_and_f_stages_vs_reference_transform_combos(dev, golden):
    """The P (marginals) and F (CDF / np.interp) stages of attwarp_mask_chain_ragged -- launched alone on a batch whose
    up-sampled mask is the fixture's uint8 attention map -- against the float32 maps the REFERENCE handed to cv2.remap for every
    uint8 combination of maps_from_attention.npz (2 maps x 6 transform names x inverse x 2 output sizes = 48 combos; the other
    48 are float attention maps, not this chain's dtype): identity / square / sqrt / unknown bit for bit, exp / log within one
    float32 ulp (device libm) -- and bit-identical to the stand-alone attwarp_axis_maps_from_attention on every combination."""
    from attwarp_amd import pipeline, new_method as nm
    g = golden("maps_from_attention")
    n_elem = n_diff = n_combo = 0
    for key in [str(c) for c in g["combos"]]:
        aname, tr, inv, wh, es, ed = key.split("|")
        att = g[aname]
        if att.dtype != np.uint8:
            continue
        n_combo += 1
        nw, nh = (int(v) for v in wh.split("x"))
        h, w = att.shape
        grid = 24 if min(h, w) > 24 else 16                 # (att_zero is 24 x 40: the ragged plan wants H, W > g)
        rb = pipeline.RaggedBatch([torch.zeros(h, w, 3, device=dev, dtype=torch.uint8)], (nh, nw), grid)
        rb.mota_of(0).copy_(T(att, dev))
        kw = dict(transform=tr, exp_scale=float(es), exp_divisor=float(ed), apply_inverse=bool(int(inv)))
        pipeline.ragged_chain_launch(P=rb, **kw)
        pipeline.ragged_chain_launch(F=rb, **kw)
        sx, sy = nm.attention_axis_maps(T(att, dev)[None], nw, nh, tr, float(es), float(ed), bool(int(inv)))
        assert torch.equal(rb.map_x, sx) or (torch.isnan(sx).any() and np.array_equal(N(rb.map_x), N(sx), equal_nan=True)), key
        assert torch.equal(rb.map_y, sy) or (torch.isnan(sy).any() and np.array_equal(N(rb.map_y), N(sy), equal_nan=True)), key
        for got, ref in ((N(rb.map_x)[0], g[f"mx|{key}"]), (N(rb.map_y)[0], g[f"my|{key}"])):
            if tr in ("identity", "square", "sqrt", "bogus"):
                assert np.array_equal(got, ref, equal_nan=True), key
                continue
            fin = np.isfinite(ref)
            assert np.array_equal(np.isnan(got), np.isnan(ref)), key
            n_elem += fin.sum(); n_diff += (got[fin] != ref[fin]).sum()
            np.testing.assert_allclose(got[fin], ref[fin], rtol=2.5e-7, atol=1e-30, err_msg=key)
    assert n_combo == 48 and n_diff <= 1e-3 * max(n_elem, 1), (n_combo, n_diff, n_elem)


def test_main_batched_loop_ragged_vs_reference_fixture(dev, golden):
    """The composed loop of the reference's batched driver on differently sized images (main_batched.py:243-287 -> llava.py:
    240-253 -> new_method.py:415-488), pinned to what the REFERENCE ITSELF produced (tests/golden/main_batched_loop.npz: five
    images, 683 x 1024 .. 640 x 427, the last with the constant 1 / 576 map whose revised mask is NaN):
      * warp_from_masks_ragged's up-sampled masks against the reference's uint8 `mota`: at most one grey level apart, the number
        of differing cells COUNTED and bounded (device exp in revise_mask is 1 ulp from torch's; x255 truncates), zero for the
        NaN map (NaN -> 0, as the reference's .byte());
      * GIVEN the reference's mota, the P and F stages of the ragged kernel reproduce the two float32 maps the reference handed
        to cv2.remap bit for bit, and the R stage the oracle's pixels on those maps bit for bit;
      * the composed output differs from that only where a mask cell flipped."""
    from attwarp_amd import pipeline
    g = golden("main_batched_loop")
    imgs, atts = main_batched_loop_inputs()
    B = len(imgs)
    d_imgs = pipeline.upload_images(imgs, dev)
    att = T(np.stack(atts), dev)
    out, rb = pipeline.warp_from_masks_ragged(d_imgs, att, (500, 500), return_batch=True)
    out = N(out)
    flips, cells = 0, 0
    for b in range(B):
        ref, got = g[f"mota_{b}"], N(rb.mota_of(b))
        d = np.abs(got.astype(np.int16) - ref.astype(np.int16))
        assert d.max() <= 1, (b, int(d.max()))
        flips += int((d != 0).sum()); cells += d.size
    print(f"main_batched loop vs the reference fixture: {flips} of {cells} up-sampled mask cells differ by one grey level")
    assert flips <= 2e-3 * cells, (flips, cells)
    assert not N(rb.mota_of(B - 1)).any()                      # the NaN mask of the constant map: zeros, as the reference's cast
    composed_maps = (N(rb.map_x).copy(), N(rb.map_y).copy())
    # given the reference's masks: maps bit for bit, pixels = the oracle's resample on the reference's maps
    for b in range(B):
        rb.mota_of(b).copy_(T(g[f"mota_{b}"], dev))
    pipeline.ragged_chain_launch(P=rb); pipeline.ragged_chain_launch(F=rb); pipeline.ragged_chain_launch(R=rb)
    mx, my, px = N(rb.map_x), N(rb.map_y), N(rb.out)
    for b in range(B):
        assert np.array_equal(mx[b], g[f"mx_{b}"]) and np.array_equal(my[b], g[f"my_{b}"]), b
        assert np.array_equal(px[b], O.remap_bilinear(imgs[b], g[f"mx_{b}"], g[f"my_{b}"], "cv2")), b
    if flips == 0:
        assert np.array_equal(out, px) and np.array_equal(composed_maps[0], mx) and np.array_equal(composed_maps[1], my)
    else:                                                       # a flipped mask cell moves the maps by ~1e-6 of a pixel
        assert np.abs(composed_maps[0] - mx).max() < 1e-2 and np.abs(out.astype(int) - px.astype(int)).max() <= 2


@pytest.mark.parametrize("n", [1, 3, 5, 6])
def test_ragged_pattern_graphs_replay_more_than_once(dev, n):
    """MaskChainStream(pattern="ragged") builds every (ring slot, batch parity) RaggedBatch in its constructor -- n of them
    for an even ring, 2 n for an odd one -- so that no table upload is ever captured into a HIP graph (a captured
    upload replays from a staging buffer that has been reused, and the resample then reads another slot's images).  Rings of
    1, 3, 5 and 6 slots with DIFFERENT images per slot: three turns of lcm(2, n) steps as graphs of several sizes (every
    graph replayed at least three times), then reset() and a second pass over fresh contents of the same buffers; every output
    equals warp_from_masks on what its slot holds.  The batches of one parity share ONE set of intermediates."""
    import math
    from attwarp_amd import pipeline
    g = torch.Generator(device=dev).manual_seed(100 + n)
    B, H, W, Ho, Wo = 3, 50, 61, 44, 52
    imgs = [torch.randint(0, 256, (B, H, W, 3), device=dev, dtype=torch.uint8, generator=g) for _ in range(n)]
    msk = [torch.rand(B, 24, 24, device=dev, generator=g) for _ in range(n)]
    mc = pipeline.MaskChainStream(imgs, msk, (Ho, Wo), pattern="ragged")
    period = math.lcm(2, n)
    assert len(mc._rb) == period
    assert len({id(rb.mota) for rb in mc._rb.values()}) == min(2, period) and len({id(rb.table_dev) for rb in mc._rb.values()}) == period
    for unroll in (2, 8):
        for o in mc.outs:
            o.zero_()
        mc.reset(); mc.prime(); mc.run(3 * period + 1, unroll=unroll); mc.drain()
        for j in range(n):
            assert torch.equal(mc.outs[j], pipeline.warp_from_masks(imgs[j], msk[j], (Ho, Wo))), (unroll, j)
    # second pass: the same buffers with new contents, the graphs captured above replayed again
    for t in imgs:
        t.copy_(torch.randint(0, 256, t.shape, device=dev, dtype=torch.uint8, generator=g))
    for m in msk:
        m.copy_(torch.rand(m.shape, device=dev, generator=g))
    mc.reset(); mc.prime(); mc.run(2 * period + 1, unroll=2); mc.drain()
    for j in range(n):
        assert torch.equal(mc.outs[j], pipeline.warp_from_masks(imgs[j], msk[j], (Ho, Wo))), j
    # a table upload inside a capture is refused, not recorded
    gr = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with pytest.raises(RuntimeError, match="capture"):
            with torch.cuda.graph(gr, stream=side):
                pipeline.RaggedBatch([imgs[0][0]], (Ho, Wo))
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()


@pytest.mark.parametrize("case", [dict(S=22, So=33, Ho=47, transform="identity"),          # sides below the 24-pixel mask grid: no staged mask up-sampling
                                  dict(S=24, So=64, Ho=64, transform="identity"),           # no up-sampling of the mask
                                  dict(S=96, So=80, Ho=80, transform="identity", mode="exact")])
def test_mask_chain_stream_falls_back_to_branches(dev, case):
    """Shapes / options attwarp_mask_chain_step refuses (ATTWARP_E_UNSUPPORTED) take the graph-branch pattern of
    MaskChainStream -- the stand-alone kernels -- and still equal warp_from_masks batch by batch; asking for "fused"
    explicitly raises."""
    from attwarp_amd import pipeline
    S, So, Ho, tr, mode = case["S"], case["So"], case["Ho"], case["transform"], case.get("mode", "cv2")
    g = torch.Generator(device=dev).manual_seed(S + So)
    n, B = 4, 3
    imgs = [torch.randint(0, 256, (B, S, S, 3), device=dev, dtype=torch.uint8, generator=g) for _ in range(n)]
    msk = [torch.rand(B, 24, 24, device=dev, generator=g) for _ in range(n)]
    mc = pipeline.MaskChainStream(imgs, msk, (Ho, So), transform=tr, mode=mode)
    assert mc.pattern == "branches" and mc.depth == 2
    mc.prime(); mc.run(6); mc.drain()
    for j in range(n):
        assert torch.equal(mc.outs[j], pipeline.warp_from_masks(imgs[j], msk[j], (Ho, So), transform=tr, mode=mode)), j
    with pytest.raises(_lib.AttWarpError):
        pipeline.MaskChainStream(imgs, msk, (Ho, So), transform=tr, mode=mode, pattern="fused")


@pytest.mark.parametrize("H,W,C,Ho,Wo,ks,coe", [(64, 96, 3, 80, 60, 3, 10.0), (96, 64, 1, 50, 100, 5, 4.0), (48, 80, 4, 64, 64, 7, 10.0),
                                                 (128, 132, 2, 90, 34, 1, 25.0), (500, 500, 3, 336, 336, 3, 10.0)])
def test_mask_chain_step_shapes_channels_filters(dev, H, W, C, Ho, Wo, ks, coe):
    """The one-launch chain step on non-square images, 1 / 2 / 3 / 4 channels, other box-filter sizes and enhancement
    strengths, down- and up-sampling outputs: every batch equals warp_from_masks with the same options."""
    from attwarp_amd import pipeline
    g = torch.Generator(device=dev).manual_seed(H + W + C)
    n, B = 6, 2
    imgs = [torch.randint(0, 256, (B, H, W, C), device=dev, dtype=torch.uint8, generator=g) for _ in range(n)]
    msk = [torch.rand(B, 24, 24, device=dev, generator=g) ** 2 for _ in range(n)]
    mc = pipeline.MaskChainStream(imgs, msk, (Ho, Wo), enhance_coe=coe, kernel_size=ks, pattern="fused")
    mc.prime(); mc.run(n - mc.depth); mc.drain()
    for j in range(n):
        ref = pipeline.warp_from_masks(imgs[j], msk[j], (Ho, Wo), enhance_coe=coe, kernel_size=ks)
        assert torch.equal(mc.outs[j], ref), j


def test_mask_chain_two_batches_per_launch(dev):
    """pipeline.pair_slots: slots 2i, 2i+1 of rings allocated as one tensor, run as ONE batch of 2B through MaskChainStream
    (two batches of the stream per launch), give the bytes of the one-batch-per-launch stream and of warp_from_masks."""
    from attwarp_amd import pipeline
    g = torch.Generator(device=dev).manual_seed(8)
    n, B, S, So = 12, 3, 96, 80
    imgs = torch.randint(0, 256, (n, B, S, S, 3), device=dev, dtype=torch.uint8, generator=g)
    msk = torch.rand(n, B, 24, 24, device=dev, generator=g)
    mc2 = pipeline.MaskChainStream(pipeline.pair_slots(list(imgs)), pipeline.pair_slots(list(msk)), (So, So))
    assert mc2.pattern == "fused" and mc2.B == 2 * B and mc2.n == n // 2
    mc2.prime(); mc2.run(n // 2 - mc2.depth); mc2.drain()
    for i in range(n):
        assert torch.equal(mc2.outs[i // 2][(i % 2) * B:(i % 2 + 1) * B], pipeline.warp_from_masks(imgs[i], msk[i], (So, So))), i
    with pytest.raises(ValueError):
        pipeline.pair_slots([imgs[0], imgs[2]])                    # not adjacent
    with pytest.raises(ValueError):
        pipeline.pair_slots(list(imgs[:3]))                        # odd


# =============================== unaligned rows and the ragged main_batched chain (round 5) ==============================
UNALIGNED_SHAPES = [(40, 683, 33, 500, 3), (37, 333, 29, 333, 3), (26, 1023, 31, 1023, 3), (30, 501, 28, 501, 3),
                    (33, 47, 40, 31, 1), (31, 29, 31, 29, 2), (28, 1365, 17, 1365, 3), (25, 4093, 9, 4095, 1), (64, 66, 50, 333, 4)]


@pytest.mark.parametrize("shape", UNALIGNED_SHAPES)
@pytest.mark.parametrize("kind", ["cdf", "wild"])
def test_remap_uint8_unaligned_rows_stay_on_the_staged_kernel(dev, shape, kind):
    """uint8 images whose rows are not a multiple of 4 bytes (683 x 3: the portrait TextVQA case) run on
    the integer cv2 kernel's unaligned form, not on the gather kernel: `remap_variant=3` (tuning flavour) REFUSES a request
    that would fall back, so a pass here proves which kernel ran.  Interleaved and planar, views that start at an odd byte,
    every prefetch depth: bit-exact against the oracle and the gather kernel."""
    from attwarp_amd import checkpoint_utils as cu
    H, W, Ho, Wo, C = shape
    rng = np.random.default_rng(H * 17 + W + Wo)
    B = 3
    mx, my = make_maps(rng, B, H, W, Ho, Wo, kind)
    img = rng.integers(0, 256, (B, H, W, C), dtype=np.uint8)
    ref = np.stack([O.remap_bilinear(img[b], mx[b], my[b], "cv2") for b in range(B)])
    ti, tx, ty = T(img, dev), T(mx, dev), T(my, dev)
    tc = T(img.transpose(0, 3, 1, 2), dev)
    with _lib.debug_override(remap_variant=3):
        hwc = N(cu.remap_separable(ti, tx, ty, mode="cv2", channels_last=True))
        chw = N(cu.remap_separable(tc, tx, ty, mode="cv2"))
    assert np.array_equal(hwc, ref) and np.array_equal(chw.transpose(0, 2, 3, 1), ref)
    with _lib.debug_override(remap_variant=1):
        assert np.array_equal(N(cu.remap_separable(ti, tx, ty, mode="cv2", channels_last=True)), ref)
    for ahead in (1, 2, 4):
        for rows in (1, 3, 7, 64):
            with _lib.debug_override(remap_variant=3, u8_ahead=ahead, remap_rows=rows):
                assert np.array_equal(N(cu.remap_separable(ti, tx, ty, mode="cv2", channels_last=True)), ref), (ahead, rows)
    # a view that starts at an odd byte of its allocation, written into a destination that does too
    flat = torch.zeros(img.size + 7, device=dev, dtype=torch.uint8)
    flat[3:3 + img.size] = ti.reshape(-1)
    src = flat[3:3 + img.size].view(B, H, W, C)
    dflat = torch.full((ref.size + 9,), 0xEE, device=dev, dtype=torch.uint8)
    dst = dflat[5:5 + ref.size].view(B, Ho, Wo, C)
    with _lib.debug_override(remap_variant=3):
        cu.remap_separable(src, tx, ty, mode="cv2", channels_last=True, out=dst)
    assert np.array_equal(N(dst), ref)
    assert bool((dflat[:5] == 0xEE).all()) and bool((dflat[5 + ref.size:] == 0xEE).all())       # nothing written outside


@pytest.mark.parametrize("shape", [(40, 683, 33, 500, 3), (37, 333, 29, 333, 3), (26, 1023, 31, 1023, 3), (30, 501, 28, 501, 3),
                                   (33, 47, 40, 31, 1), (31, 29, 31, 29, 2), (28, 1365, 17, 1365, 3), (25, 4093, 9, 4095, 1),
                                   (30, 1023, 24, 1024, 3), (27, 683, 20, 1024, 3)])
@pytest.mark.parametrize("mode", ["cv2", "exact"])
def test_remap_float32_unaligned_rows_stay_on_the_staged_kernel(dev, shape, mode):
    """float32 images whose rows are not a multiple of 4 floats (W * C % 4 != 0: 683 x 3, 333 x 3, 1023 x 3, 501 x 3) run
    on remap_rows_kernel's unaligned form in both arithmetic modes -- `remap_variant=3` refuses a fall-back to the gather
    kernel -- interleaved and planar (plane by plane), every rows-per-block / row-blocks-per-workgroup setting, and from a
    view that starts 4 bytes into a 16-byte line: bit-exact against the oracle and the gather kernel."""
    from attwarp_amd import checkpoint_utils as cu
    H, W, Ho, Wo, C = shape
    rng = np.random.default_rng(H * 19 + W + Wo)
    B = 3
    for kind in ("cdf", "wild"):
        mx, my = make_maps(rng, B, H, W, Ho, Wo, kind)
        img = rng.random((B, H, W, C), dtype=np.float32)
        ref = np.stack([O.remap_bilinear(img[b], mx[b], my[b], mode) for b in range(B)])
        ti, tx, ty = T(img, dev), T(mx, dev), T(my, dev)
        tc = T(img.transpose(0, 3, 1, 2), dev)
        with _lib.debug_override(remap_variant=3):
            hwc = N(cu.remap_separable(ti, tx, ty, mode=mode, channels_last=True))
            chw = N(cu.remap_separable(tc, tx, ty, mode=mode))
        assert np.array_equal(hwc, ref), kind
        assert np.array_equal(chw.transpose(0, 2, 3, 1), ref), kind
        with _lib.debug_override(remap_variant=1):
            assert np.array_equal(N(cu.remap_separable(ti, tx, ty, mode=mode, channels_last=True)), ref)
        for rows, cpw in ((1, 1), (3, 2), (7, 3), (64, 1)):
            with _lib.debug_override(remap_variant=3, remap_rows=rows, remap_cpw=cpw):
                assert np.array_equal(N(cu.remap_separable(ti, tx, ty, mode=mode, channels_last=True)), ref), (kind, rows, cpw)
        flat = torch.zeros(img.size + 8, device=dev, dtype=torch.float32)
        flat[1:1 + img.size] = ti.reshape(-1)
        with _lib.debug_override(remap_variant=3):
            got = cu.remap_separable(flat[1:1 + img.size].view(B, H, W, C), tx, ty, mode=mode, channels_last=True)
        assert np.array_equal(N(got), ref), kind


@pytest.mark.parametrize("hw", [(64, 683), (70, 333), (129, 1023), (501, 501), (1024, 683), (90, 130), (33, 26)])
def test_mask_upsample_and_marginals_unaligned_widths(dev, hw):
    """The LANCZOS up-sampling of the 24 x 24 mask to a width that is not a multiple of 4, and the float64 marginals of
    that uint8 mask, on their staged kernels' unaligned forms (column-strip kernel / byte-packed profile kernel): the mask
    bit-exact against Pillow's arithmetic (the oracle), the maps bit-exact against numpy's summation orders, both equal to
    the generic kernels."""
    from attwarp_amd import attention_extraction as ae, new_method as nm
    H, W = hw
    rng = np.random.default_rng(H * 3 + W)
    B = 3
    rev = rng.random((B, 24, 24), dtype=np.float32)
    mota = ae.upsample_mask_lanczos(T(rev, dev), (W, H))
    with _lib.debug_override(lanczos_variant=1):
        mota2 = ae.upsample_mask_lanczos(T(rev, dev), (W, H))
    assert torch.equal(mota, mota2)
    for b in range(B):
        assert np.array_equal(N(mota[b]), O.lanczos_resize_u8(O.mask_to_u8(rev[b]), W, H)), b
    for tr in ("identity", "sqrt"):
        mx, my = nm.attention_axis_maps(mota, 500, 470, tr)
        with _lib.debug_override(profiles_variant=1):
            gx, gy = nm.attention_axis_maps(mota, 500, 470, tr)
        assert torch.equal(mx, gx) and torch.equal(my, gy)
        for b in range(B):
            omx, omy = O.maps_from_attention(N(mota[b]), 500, 470, tr)
            assert np.array_equal(N(mx[b]), omx) and np.array_equal(N(my[b]), omy), (tr, b)
    # a mask that starts at an odd byte
    flat = torch.zeros(mota.numel() + 5, device=dev, dtype=torch.uint8)
    flat[1:1 + mota.numel()] = mota.reshape(-1)
    vx, vy = nm.attention_axis_maps(flat[1:1 + mota.numel()].view(B, H, W), 500, 470, "identity")
    mx, my = nm.attention_axis_maps(mota, 500, 470, "identity")
    assert torch.equal(vx, mx) and torch.equal(vy, my)


# (W x H as PIL reports them) of a TextVQA-like batch: OpenImages landscape / portrait / square, Flickr-sized, small ones
TEXTVQA_LIKE = [(1024, 768), (683, 1024), (1024, 1024), (500, 375), (333, 500), (640, 427)]


def ragged_batch(dev, B, seed, extra=()):
    g = torch.Generator(device=dev).manual_seed(seed)
    sizes = [TEXTVQA_LIKE[(b + seed) % len(TEXTVQA_LIKE)] for b in range(B)]
    for i, wh in enumerate(extra):
        sizes[(3 * i + 1) % B] = wh
    images = [torch.randint(0, 256, (h, w, 3), device=dev, dtype=torch.uint8, generator=g) for (w, h) in sizes]
    att = torch.rand(B, 24, 24, device=dev, generator=g) ** 3
    att = att / att.sum((1, 2), keepdim=True)
    return images, att


def test_main_batched_ragged_chain_vs_oracle_every_image(dev):
    """The contract of AGW/main_batched.py:243-287 as that driver holds it: a batch of 32 images of DIFFERENT sizes
    (1024 x 768, 683 x 1024, 1024 x 1024, 500 x 375, 333 x 500, 640 x 427, plus odd ones), each with its 24 x 24 attention
    map, warped to 500 x 500 by five ragged launches (pipeline.warp_from_masks_ragged).  EVERY image against the oracle
    stage by stage, as test_main_batched_chain_full_size_vs_oracle does for equal sizes: revised mask <= 1 ulp; given it,
    the up-sampled uint8 mask (Pillow's arithmetic), the float32 maps and the uint8 pixels bit-exact; and equal to the
    per-image drop-in (pipeline.warp_from_masks on a batch of one)."""
    import subprocess
    from attwarp_amd import pipeline
    from oracle import c_oracle as C
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "oracle")], check=True, capture_output=True)
    B = 32
    images, att = ragged_batch(dev, B, 500, extra=[(1365, 31), (25, 300), (1023, 501), (612, 612)])
    att[2] = 1.0 / 576                      # a constant map: revise_mask divides 0 by 0, the uint8 mask is all zero -> the fallback branch
    att[6] = 0; att[6, 10:13, 4:7] = 1.0    # one 3 x 3 hot spot: strong magnification there, minification elsewhere
    out, rb = pipeline.warp_from_masks_ragged(images, att, (500, 500), return_batch=True)
    assert rb is not None and tuple(out.shape) == (B, 500, 500, 3)
    torch.cuda.synchronize()
    att_h, rev_h, out_h, mx_h, my_h = N(att), N(rb.rev), N(out), N(rb.map_x), N(rb.map_y)
    flips = 0
    for b in range(B):
        H, W = int(images[b].shape[0]), int(images[b].shape[1])
        with np.errstate(all="ignore"):
            orev = O.revise_mask(att_h[b], 3, 10)
        assert np.array_equal(np.isnan(rev_h[b]), np.isnan(orev)), b
        fin = np.isfinite(orev)
        u = np.abs(rev_h[b].view(np.int32).astype(np.int64) - orev.view(np.int32).astype(np.int64))
        assert not fin.any() or u[fin].max() <= 1, b
        mota = N(rb.mota_of(b))
        assert np.array_equal(mota, O.lanczos_resize_u8(O.mask_to_u8(rev_h[b]), W, H)), (b, H, W)
        omx, omy = O.maps_from_attention(mota, 500, 500, "identity")
        assert np.array_equal(mx_h[b], omx) and np.array_equal(my_h[b], omy), (b, H, W)
        assert np.array_equal(out_h[b], C.remap_bilinear_u8(N(images[b]), omx, omy, "cv2")), (b, H, W)
        one = pipeline.warp_from_masks(images[b][None], att[b:b + 1], (500, 500))
        assert torch.equal(one[0], out[b]), (b, H, W)
        with np.errstate(all="ignore"):
            d = np.abs(O.mask_to_u8(orev).astype(int) - O.mask_to_u8(rev_h[b]).astype(int))
        assert d.max() <= 1
        flips += int((d > 0).sum())
    assert flips <= 1e-3 * B * 576


def test_ragged_stream_equals_serial_launches(dev):
    """pipeline.RaggedMaskChainStream -- one launch per batch, R(k) | F(k+1) | P(k+2) | L(k+3) | V(k+4) on five different
    ragged batches -- over a stream of batches of different compositions AND different batch sizes (shorter and longer than
    the pipeline): every output equals warp_from_masks_ragged on that batch byte for byte; likewise the ring form, eager and
    as HIP graphs."""
    from attwarp_amd import pipeline
    comps = [(5, 1, [(683, 512)]), (3, 2, []), (8, 3, [(100, 77), (1365, 40)]), (1, 4, []), (6, 5, [(501, 333)]), (4, 6, []), (7, 7, [])]
    batches = [ragged_batch(dev, B, seed, extra) for (B, seed, extra) in comps]
    # output rows that are NOT a multiple of 4 bytes (319 x 3 = 957) from aligned and unaligned images alike: every image
    # against the per-image drop-in (the resample of an aligned image must take the form that stores the row's last bytes)
    for (i, a) in batches[:3]:
        odd = pipeline.warp_from_masks_ragged(i, a, (41, 319))
        for b in range(len(i)):
            assert torch.equal(odd[b], pipeline.warp_from_masks(i[b][None], a[b:b + 1], (41, 319))[0]), b
    want = [pipeline.warp_from_masks_ragged(i, a, (120, 136)) for (i, a) in batches]
    for n in (1, 3, 4, 5, 7):
        st = pipeline.RaggedMaskChainStream(out_size=(120, 136))
        got = []
        for (i, a) in batches[:n]:
            d = st.push(i, a)
            if d is not None:
                got.append(d.out)
        got += [d.out for d in st.flush()]
        assert len(got) == n and all(torch.equal(g, w) for g, w in zip(got, want)), n
    # ring form: 6 prebuilt batches, 2 turns; eager and as graphs
    for unroll in (0, 6):
        st = pipeline.RaggedMaskChainStream(out_size=(120, 136))
        ring = []
        for (i, a) in batches[:6]:
            rb = pipeline.RaggedBatch(i, (120, 136)); rb.masks = a.float().contiguous()
            ring.append(rb)
        st.ring(ring)
        st.prime()
        st.run(12, unroll=unroll)
        st.drain_ring()
        torch.cuda.synchronize()
        assert all(torch.equal(rb.out, w) for rb, w in zip(ring, want)), unroll


def test_remap_kernels_vs_real_opencv_fixture_if_present(dev):
    """tests/golden/remap_cv2.npz = REAL cv2.remap outputs (written by `make_golden.py --with-opencv` on a machine that has
    OpenCV; absent until then: the cv2 arithmetic is pinned to OpenCV's published algorithm only).  When it is
    there, every resample kernel family must reproduce it bit for bit."""
    from attwarp_amd import checkpoint_utils as cu
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "remap_cv2.npz")
    if not os.path.exists(path):
        pytest.skip("no remap_cv2.npz: OpenCV has not been available to this repository yet")
    g = np.load(path, allow_pickle=False)
    for key in [str(k) for k in g["cases"]]:
        img, mx, my, want = g[f"{key}|img"], g[f"{key}|mx"], g[f"{key}|my"], g[f"{key}|out"]
        x = img[None, :, :, None] if img.ndim == 2 else img[None]
        for variant in (-1, 1):
            with _lib.debug_override(remap_variant=variant):
                got = N(cu.remap_separable(T(x, dev), T(mx[None], dev), T(my[None], dev), mode="cv2", channels_last=True))[0]
            assert np.array_equal(got.reshape(want.shape), want), (key, variant, str(g["opencv_build"]))


def test_upload_images_packs_host_images_for_the_ragged_chain(dev):
    """pipeline.upload_images: PIL images / arrays of different sizes -> one staging buffer, one transfer, GPU views at
    whatever byte offsets the packing gives (odd ones included) -> the ragged chain; equals the chain on separately
    allocated copies."""
    from PIL import Image
    from attwarp_amd import pipeline
    rng = np.random.default_rng(91)
    sizes = [(75, 101), (64, 99), (33, 131), (90, 64)]
    host = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for (h, w) in sizes]
    mixed = [Image.fromarray(host[0]), host[1], Image.fromarray(host[2]), host[3]]
    up = pipeline.upload_images(mixed, dev)
    assert [tuple(u.shape) for u in up] == [(h, w, 3) for (h, w) in sizes]
    assert any(u.data_ptr() % 4 for u in up)                         # the packing really produces unaligned images
    assert all(np.array_equal(N(u), h) for u, h in zip(up, host))
    att = torch.rand(4, 24, 24, device=dev)
    a = pipeline.warp_from_masks_ragged(up, att, (60, 72))
    b = pipeline.warp_from_masks_ragged([T(h, dev) for h in host], att, (60, 72))
    assert torch.equal(a, b)


def test_ragged_large_batch_of_small_images(dev):
    """A ragged batch of 1500 small images (more images than the chip has workgroup slots, more than 8 bits of image index,
    a batch size that is not a multiple of 8: the resample's plain block order): a sample of images against the per-image
    drop-in, the whole output against a second run and against the stream form."""
    from attwarp_amd import pipeline
    g = torch.Generator(device=dev).manual_seed(1500)
    B = 1500
    hs = torch.randint(25, 70, (B,), generator=torch.Generator().manual_seed(1)).tolist()
    ws = torch.randint(25, 70, (B,), generator=torch.Generator().manual_seed(2)).tolist()
    images = [torch.randint(0, 256, (h, w, 3), device=dev, dtype=torch.uint8, generator=g) for h, w in zip(hs, ws)]
    att = torch.rand(B, 24, 24, device=dev, generator=g) ** 2
    out = pipeline.warp_from_masks_ragged(images, att, (48, 52))
    assert torch.equal(out, pipeline.warp_from_masks_ragged(images, att, (48, 52)))
    for b in list(range(0, B, 97)) + [255, 256, 257, B - 1]:
        assert torch.equal(out[b], pipeline.warp_from_masks(images[b][None], att[b:b + 1], (48, 52))[0]), b
    st = pipeline.RaggedMaskChainStream(out_size=(48, 52))
    got = [st.push(images, att) for _ in range(5)]
    got = [d for d in got if d is not None] + st.flush()
    assert len(got) == 5 and all(torch.equal(d.out, out) for d in got)


def test_ragged_falls_back_per_image_outside_its_limits(dev):
    """Images the ragged kernel does not take (rows wider than 4096 bytes, a side <= 24) run through warp_from_masks one by
    one inside warp_from_masks_ragged; the rest of the batch still runs ragged; results are those of the per-image path."""
    from attwarp_amd import pipeline
    g = torch.Generator(device=dev).manual_seed(77)
    sizes = [(40, 1500), (64, 100), (24, 90), (90, 77)]
    images = [torch.randint(0, 256, (h, w, 3), device=dev, dtype=torch.uint8, generator=g) for (h, w) in sizes]
    att = torch.rand(4, 24, 24, device=dev, generator=g)
    out = pipeline.warp_from_masks_ragged(images, att, (60, 72))
    for b in range(4):
        assert torch.equal(out[b], pipeline.warp_from_masks(images[b][None], att[b:b + 1], (60, 72))[0]), b
    with pytest.raises(_lib.AttWarpError):
        pipeline.RaggedBatch(images, (60, 72))
    # an axis so long that the finalize body's LDS does not fit: ragged_eligible says yes, the plan says no -> per image
    tall = [torch.randint(0, 256, (7600, 40, 3), device=dev, dtype=torch.uint8, generator=g), images[1]]
    assert pipeline.ragged_eligible(7600, 40, 3)
    out = pipeline.warp_from_masks_ragged(tall, att[:2], (60, 72))
    for b in range(2):
        assert torch.equal(out[b], pipeline.warp_from_masks(tall[b][None], att[b:b + 1], (60, 72))[0]), b


def test_randomised_differential_runs(dev):
    """A short run of the two fuzzers (tests/fuzz/fuzz_remap.py, tests/fuzz/fuzz_stages.py: random shapes, dtypes, layouts, modes,
    hostile values; every stage entry point against the oracle): no mismatch.  The long runs behind DESIGN section 4 are
    the same scripts with a larger time budget."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for script, args in (("fuzz_remap.py", ["6", "11"]), ("fuzz_stages.py", ["1", "11"])):
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz", script)] + args, capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "MISMATCH" not in r.stdout and "EXCEPTION" not in r.stdout, r.stdout[-3000:]
        lines = [l for l in r.stdout.splitlines() if "mismatches" in l]
        assert lines and all(" 0 mismatches" in l for l in lines), r.stdout[-3000:]

