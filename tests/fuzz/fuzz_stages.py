"""Test infrastructure (imports oracle/): randomised differential run of the per-stage entry points against the numpy oracle (bit for bit unless a
tolerance is written next to the stage): attention maps from uint8 / float32 / float64 attention with every transform
(A13), PDF -> CDF -> maps chain with hostile densities (A8-A11), cdf repair / resample (A10), attention reduce with
random geometry (A1/A2), LANCZOS mask up-sample vs the Pillow restatement (A4), revise_mask (A3), adaptive pool and
gt_marginals (A5/A6), safe_softmax (A7).   usage: fuzz_stages.py [seconds per stage] [seed]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from attwarp_amd import checkpoint_utils as cu, attention_extraction as ae, new_method as nm, pipeline, model
from oracle import warp_oracle as O
dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
N = lambda t: t.detach().cpu().numpy()
def run(name, gen):
    t0 = time.time(); n = bad = 0
    while time.time() - t0 < budget:
        with np.errstate(all="ignore"):
            r = gen()
        if r is None: continue
        ok, desc = r
        n += 1
        if not ok:
            bad += 1
            if bad <= 5: print("  MISMATCH", name, desc, flush=True)
    print(f"{name:34s} {n:6d} cases  {bad:5d} mismatches", flush=True)

def dim(lo=1, hi=1300):
    r = rng.random()
    if r < 0.4: return int(rng.integers(lo, 64))
    if r < 0.8: return int(rng.integers(64, 400))
    return int(rng.integers(400, hi))

def g_att_maps():
    B = int(rng.integers(1, 4)); h, w = dim(2), dim(2)
    kind = rng.choice(["u8", "f32", "f64"])
    if kind == "u8": att = rng.integers(0, 256, (B, h, w), dtype=np.uint8)
    elif kind == "f32": att = (rng.random((B, h, w), dtype=np.float32) * 4 - 0.5).astype(np.float32)
    else: att = rng.random((B, h, w)) * 4 - 0.5
    r = rng.random()
    if r < 0.15: att[0] = 0
    elif r < 0.3: att[0, : h // 2] = 0
    elif r < 0.4 and kind != "u8": att[0, rng.integers(0, h), rng.integers(0, w)] = rng.choice([np.nan, np.inf, 1e30, -5.0])
    tr = rng.choice(["identity", "square", "sqrt", "exp", "log"]); inv = bool(rng.random() < 0.3)
    nw, nh = dim(1, 1100), dim(1, 1100)
    es, ed = float(rng.choice([1.0, 0.05, 3.0])), float(rng.choice([1.0, 50.0]))
    mx, my = nm.attention_axis_maps(T(att), nw, nh, tr, es, ed, inv)
    ok = True
    if tr in ("identity", "square") and not inv:        # numpy's own summation orders: bit for bit
        for b in range(B):
            rx, ry = O.maps_from_attention(att[b], nw, nh, tr, es, ed, inv)
            ok &= np.array_equal(N(mx)[b], rx, equal_nan=True) and np.array_equal(N(my)[b], ry, equal_nan=True)
    else:                                               # device libm vs numpy libm: 1 ulp on exp / log / sqrt sums
        for b in range(B):
            rx, ry = O.maps_from_attention(att[b], nw, nh, tr, es, ed, inv)
            for g, r_ in ((N(mx)[b], rx), (N(my)[b], ry)):
                fin = np.isfinite(r_) & np.isfinite(g)
                ok &= np.array_equal(np.isfinite(r_), np.isfinite(g)) and (not fin.any() or np.max(np.abs(g[fin] - r_[fin])) <= 2e-3 * max(1.0, np.max(np.abs(r_[fin]))) * 1e-3 + 1e-3)
    return ok, (kind, B, h, w, tr, inv, nw, nh, es, ed)

def g_pdf_chain():
    B = int(rng.integers(1, 5)); W, H = dim(24), dim(24); Wo, Ho = dim(1), dim(1)
    px = rng.random((B, 24), dtype=np.float32); py = rng.random((B, 24), dtype=np.float32)
    r = rng.random()
    if r < 0.2: px[0] = 0
    elif r < 0.3: px[0, rng.integers(0, 24)] = rng.choice([np.nan, np.inf, -1.0, 1e30])
    elif r < 0.5: px = (px ** 8).astype(np.float32)
    px /= np.maximum(px.sum(1, keepdims=True), 1e-6); py /= py.sum(1, keepdims=True)
    mx, my = pipeline.axis_maps_from_pdf(T(px), T(py), (H, W), (Ho, Wo))
    Fx = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(px, W), 0))
    Fy = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(py, H), 0))
    rx, ry = O.maps_from_cdf(Fx, Fy, (Ho, Wo))
    ok = np.array_equal(N(mx), rx, equal_nan=True) and np.array_equal(N(my), ry, equal_nan=True)      # bit for bit
    return ok, (B, W, H, Wo, Ho)

def g_cdf_stages():
    B = int(rng.integers(1, 4)); L = dim(2); L2 = dim(2)
    p = rng.random((B, L), dtype=np.float32)
    r = rng.random()
    if r < 0.2: p[0] = 0
    elif r < 0.4: p[0, rng.integers(0, L)] = rng.choice([np.nan, np.inf, -3.0])
    elif r < 0.6: p = (p ** 12).astype(np.float32)
    F = cu.cdf_from_density(T(p)); Fo = O.cdf_from_density(p)
    ok = np.array_equal(N(F), Fo, equal_nan=True)
    G = rng.random((B, L), dtype=np.float32); G.sort(axis=1)
    if rng.random() < 0.5: G[:, L // 3: L // 2] = G[:, L // 3: L // 3 + 1]
    ok &= np.array_equal(N(cu._make_strictly_increasing(T(G))), O.make_strictly_increasing(G), equal_nan=True)
    ok &= np.allclose(N(cu.resample_cdf(T(G), L2)), O.resample_cdf(G, L2), rtol=3e-7, atol=1e-7, equal_nan=True)
    return ok, (B, L, L2)

def g_attn():
    B = int(rng.integers(1, 5)); heads = int(rng.integers(1, 40)); ntok = int(rng.choice([576, 576, 64, 100, 577, 4, 1000]))
    kv = ntok + int(rng.integers(0, 90)); q = int(rng.integers(1, 3))
    dt = rng.choice([np.float32, np.float16])
    a = rng.random((B, heads, q, kv), dtype=np.float32)
    if rng.random() < 0.3: a = np.exp(rng.normal(0, 5, a.shape)).astype(np.float32); a /= a.sum(-1, keepdims=True)
    a = a.astype(dt)
    starts = [int(rng.integers(0, kv - ntok + 1)) for _ in range(B)]
    ends = [s + ntok for s in starts]
    hl = ae.BatchMaskHookLogger(model=None, device=dev)
    hl.set_batch_image_token_ranges(starts, ends)
    hl._process_attention(T(a))
    ok = np.array_equal(N(hl.step_attentions[-1]), O.attn_reduce_step(a, starts, ends), equal_nan=True)
    return ok, (B, heads, ntok, kv, q, dt.__name__, starts)

def g_lanczos():
    B = int(rng.integers(1, 4)); h, w = int(rng.integers(2, 40)), int(rng.integers(2, 40))
    H, W = dim(1, 1200), dim(1, 1200)
    m = rng.random((B, h, w), dtype=np.float32)
    up = ae.upsample_mask_lanczos(T(m), (W, H))
    ok = all(np.array_equal(N(up)[b], O.lanczos_resize_u8(O.mask_to_u8(m[b]), W, H)) for b in range(B))
    return ok, (B, h, w, H, W)

def g_pool_marg():
    B = int(rng.integers(1, 3)); H, W = dim(24), dim(24)
    A = (rng.random((B, 1, H, W), dtype=np.float32) * 3 - 0.5).astype(np.float32)
    ok = np.array_equal(N(pipeline.adaptive_avg_pool2d(T(A))), O.adaptive_avg_pool24(A))
    px, py = cu.gt_marginals(T(A)); pxo, pyo = O.gt_marginals(A)
    ok &= np.array_equal(N(px), pxo) and np.array_equal(N(py), pyo)
    return ok, (B, H, W)

def g_revise_softmax():
    B = int(rng.integers(1, 5)); n = int(rng.choice([24, 24, 16, 7, 32])); ks = int(rng.choice([1, 3, 5, 7])); coe = float(rng.choice([10.0, 1.0, 30.0]))
    m = rng.random((B, n, n), dtype=np.float32)
    if rng.random() < 0.2: m[0] = m[0, 0, 0]
    got = N(ae.revise_mask(T(m), kernel_size=ks, enhance_coe=coe))
    ref = np.stack([O.revise_mask(m[b], ks, coe) for b in range(B)])
    ok = np.allclose(got, ref, rtol=0, atol=3e-7, equal_nan=True)              # device expf vs numpy: <= 1 ulp before the box filter
    L = dim(2); x = (rng.standard_normal((B, L)) * 10).astype(np.float32)
    if rng.random() < 0.4: x[0, rng.integers(0, L)] = rng.choice([np.nan, np.inf, -np.inf])
    ok &= np.allclose(N(model.safe_softmax(T(x))), O.safe_softmax(x), rtol=3e-7, atol=1e-9)
    return ok, (B, n, ks, coe, L)

def g_clip():
    B = int(rng.integers(1, 3)); H, W = int(rng.integers(8, 700)), int(rng.integers(8, 700)); size = int(rng.choice([336, 224, 335, 64, 100]))
    C = 3
    img = rng.integers(0, 256, (B, H, W, C), dtype=np.uint8)
    pad = bool(rng.random() < 0.3)
    got = N(pipeline.clip_preprocess(T(img), size, torch.float32, pad_to_square=pad))
    ok = True
    for b in range(B):
        src = O.expand2square(img[b]) if pad else img[b]
        ok &= np.array_equal(got[b], O.clip_preprocess(src, size))
    return ok, (B, H, W, size, pad)

def g_probe():
    B = int(rng.integers(1, 4)); D = int(rng.choice([128, 64, 96])); Hkv = int(rng.choice([1, 2, 4, 8])); H = Hkv * int(rng.choice([1, 2, 4]))
    ntok = int(rng.choice([576, 64, 100])); kv = ntok + int(rng.integers(8, 120))
    dt = rng.choice([np.float16, np.float32])
    q = rng.standard_normal((B, H, D)).astype(dt); k = rng.standard_normal((B, Hkv, kv, D)).astype(dt)
    pads = np.array([int(rng.integers(0, 8)) for _ in range(B)], np.int32)
    starts = np.array([int(rng.integers(int(pads[b]), kv - ntok + 1)) for b in range(B)], np.int32)
    sc = float(D) ** -0.5
    got = N(ae.probe_last_query(T(q), T(k), T(starts), ntok, T(pads), sc))
    ok = np.array_equal(got, O.attn_probe_step(q, k, starts, ntok, sc, pads), equal_nan=True)
    return ok, (B, H, Hkv, D, ntok, kv, dt.__name__)

def g_mn_tail():
    B = int(rng.integers(1, 5)); Lt = int(rng.integers(1, 40)); D = int(rng.choice([16, 100, 4096])); Ch = int(rng.choice([4, 16, 256])); Hh, Ww = int(rng.integers(1, 30)), int(rng.integers(1, 30))
    tok = rng.standard_normal((B, Lt, D)).astype(np.float32); msk = (rng.random((B, Lt)) > 0.4).astype(np.float32)
    if rng.random() < 0.2: msk[0] = 0
    ok = np.array_equal(N(model.masked_token_mean(T(tok), T(msk[:, :, None]))), O.masked_token_mean(tok, msk))
    v = rng.standard_normal((B, Ch, Hh, Ww)).astype(np.float32); gb = rng.standard_normal((B, 2 * Ch)).astype(np.float32)
    vx, vy = model.film_axis_means(T(v), T(gb)); ox, oy = O.film_axis_means(v, gb)
    ok &= np.array_equal(N(vx), ox) and np.array_equal(N(vy), oy)
    return ok, (B, Lt, D, Ch, Hh, Ww)

def g_marginalnet():
    dv, dtx, hid = int(rng.integers(4, 40)), int(rng.integers(4, 40)), int(rng.integers(2, 24))
    B, Lt = int(rng.integers(1, 4)), int(rng.integers(1, 9))
    fh, fw = int(rng.integers(3, 26)), int(rng.integers(3, 26)); H, W = (fh, fw) if rng.random() < 0.5 else (int(rng.integers(3, 30)), int(rng.integers(3, 30)))
    torch.manual_seed(int(rng.integers(0, 1 << 30)))
    net = model.MarginalNet(dv, dtx, hid).to(dev).eval()
    fmap = torch.randn(B, dv, fh, fw, device=dev); tok = torch.randn(B, Lt, dtx, device=dev); msk = (torch.rand(B, Lt, 1, device=dev) > 0.4).float()
    with torch.no_grad():
        px, py = net(fmap, H, W, tok, msk)                                  # library GEMMs + the fused HIP tail
        lx, ly = net.forward_logits(fmap, H, W, tok, msk)                   # all stock ops
    ok = np.allclose(N(px), O.safe_softmax(N(lx)), rtol=3e-6, atol=1e-9) and np.allclose(N(py), O.safe_softmax(N(ly)), rtol=3e-6, atol=1e-9)
    return ok, (dv, dtx, hid, B, Lt, fh, fw, H, W)

def g_stack_chain():
    """attention stack -> warped batch (the bench's path) against the oracle stage by stage, every size a multiple of
    nothing in particular; float32 and float16 rows, both layouts and modes."""
    B = int(rng.integers(1, 4)); Tn = int(rng.integers(1, 6)); heads = int(rng.integers(1, 9)); kv = 576 + int(rng.integers(0, 70))
    H, W = int(rng.integers(24, 120)), int(rng.integers(24, 120))
    osz = None if rng.random() < 0.5 else (int(rng.integers(4, 130)), int(rng.integers(4, 130)))
    dt = np.float32 if rng.random() < 0.6 else np.float16
    rows = rng.random((Tn, B, heads, kv), dtype=np.float32)
    if rng.random() < 0.3: rows = np.exp(rng.normal(0, 3, rows.shape)).astype(np.float32)
    rows = (rows / rows.sum(-1, keepdims=True)).astype(dt)
    starts = np.array([int(rng.integers(0, kv - 576 + 1)) for _ in range(B)], np.int32)
    cl = bool(rng.random() < 0.5); mode = str(rng.choice(["cv2", "exact"]))
    img = rng.random((B, H, W, 3), dtype=np.float32)
    x = T(img if cl else img.transpose(0, 3, 1, 2))
    got = N(pipeline.warp_from_attention_stack(x, T(rows), T(starts), osz, channels_last=cl, mode=mode))
    if not cl: got = got.transpose(0, 2, 3, 1)
    att = O.attn_reduce_stack(rows, starts).astype(np.float32).reshape(B, 1, 24, 24)
    px, py = O.gt_marginals(att)
    Fx = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(px, W), 0)); Fy = O.cdf_from_density(np.maximum(O.upsample_pdf_right_inverse(py, H), 0))
    mx, my = O.maps_from_cdf(Fx, Fy, osz if osz else (H, W))
    ref = np.stack([O.remap_bilinear(img[b], mx[b], my[b], mode) for b in range(B)])
    ok = np.array_equal(got, ref)                                                                       # bit for bit
    return ok, (B, Tn, heads, kv, H, W, osz, dt.__name__, cl, mode, float(np.abs(got - ref).max()))

def g_chain_u8():
    h, w = dim(8, 700), dim(8, 700); nw, nh = dim(4, 700), dim(4, 700)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    att = rng.integers(0, 256, (h, w), dtype=np.uint8)
    if rng.random() < 0.1: att[:] = 0
    tr = rng.choice(["identity", "square"]); mode = rng.choice(["cv2", "exact"])
    got = nm.warp_image_by_attention(img, att, nw, nh, transform=tr, exp_scale=1.0, exp_divisor=1.0, apply_inverse=False, mode=mode)
    ok = np.array_equal(got, O.warp_image_by_attention(img, att, nw, nh, tr, mode=mode))
    return ok, (h, w, nw, nh, tr, mode)

def rand_transform():
    """save_warped_image's keyword arguments (new_method.py:405-411), identity half of the time (what both drivers pass)."""
    if rng.random() < 0.5:
        return {}
    tr = str(rng.choice(["identity", "square", "sqrt", "exp", "log", "bogus"]))
    es, ed = [(1.0, 1.0), (0.01, 2.0), (0.02, 1.5)][int(rng.integers(0, 3))] if tr == "exp" else (1.0, 1.0)
    return dict(transform=tr, exp_scale=es, exp_divisor=ed, apply_inverse=bool(rng.random() < 0.4))

def g_mask_chain_stream():
    """pipeline.MaskChainStream (the main_batched chain as a batch stream: the one-launch step where the shape is eligible,
    graph branches otherwise) against pipeline.warp_from_masks batch by batch, and the first batch against the oracle."""
    B = int(rng.integers(1, 5)); S = 4 * int(rng.integers(8, 180)); So = 4 * int(rng.integers(2, 180)); Ho = int(rng.integers(3, 700))
    n = int(rng.choice([5, 6, 8])); nb = n + int(rng.integers(0, 4))
    imgs = [T(rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8)) for _ in range(n)]
    msk = [T(rng.random((B, 24, 24), dtype=np.float32) ** int(rng.integers(1, 5))) for _ in range(n)]
    xf = rand_transform()
    mc = pipeline.MaskChainStream(imgs, msk, (Ho, So), **xf)
    mc.prime(); mc.run(nb - mc.depth if nb > mc.depth else 0, unroll=int(rng.choice([2, 4])));
    if nb >= mc.depth: mc.drain()
    done = min(nb, n) if nb >= mc.depth else 0
    ok = True
    for j in range(max(0, nb - n), nb if nb >= mc.depth else 0):
        ok = ok and bool(torch.equal(mc.outs[j % n], pipeline.warp_from_masks(imgs[j % n], msk[j % n], (Ho, So), **xf)))
    return ok, (B, S, So, Ho, n, nb, mc.pattern, xf)

def g_ragged_chain():
    """pipeline.warp_from_masks_ragged / RaggedMaskChainStream on batches of random sizes (unaligned widths, sides near the
    24-pixel limit, rows up to the 4096-byte limit, ineligible images mixed in): every image against the per-image drop-in
    (warp_from_masks on a batch of one, itself checked against the oracle above) and one image per batch against the oracle."""
    B = int(rng.integers(1, 9)); Ho, Wo = int(rng.integers(3, 300)), int(rng.integers(2, 400))
    def side():
        r = rng.random()
        return int(rng.integers(25, 90)) if r < 0.4 else int(rng.integers(90, 500)) if r < 0.85 else int(rng.integers(500, 1366))
    nb = int(rng.integers(1, 7))
    batches = []
    for _ in range(nb):
        sizes = [(side(), side()) for _ in range(B)]
        if rng.random() < 0.15: sizes[0] = (int(rng.integers(8, 25)), side())          # not eligible: falls back per image
        if rng.random() < 0.1: sizes[-1] = (side(), int(rng.integers(1366, 1800)))     # rows wider than 4096 bytes: likewise
        batches.append(([T(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)) for (h, w) in sizes],
                        T(rng.random((B, 24, 24), dtype=np.float32) ** int(rng.integers(1, 5)))))
    ok = True
    xf = rand_transform()
    outs = [pipeline.warp_from_masks_ragged(i, a, (Ho, Wo), **xf) for (i, a) in batches]
    for (imgs, att), out in zip(batches, outs):
        for b in range(B):
            ok = ok and bool(torch.equal(out[b], pipeline.warp_from_masks(imgs[b][None], att[b:b + 1], (Ho, Wo), **xf)[0]))
    imgs, att = batches[0]
    b = int(rng.integers(0, B)); h, w = int(imgs[b].shape[0]), int(imgs[b].shape[1])
    rev = N(ae.revise_mask(att[b:b + 1]))[0]
    if xf.get("transform", "identity") not in ("exp", "log"):          # (exp / log: the device libm may move a map entry by one ulp)
        ref = O.warp_image_by_attention(N(imgs[b]), O.lanczos_resize_u8(O.mask_to_u8(rev), w, h), Wo, Ho, xf.get("transform", "identity"),
                                        xf.get("exp_scale", 1.0), xf.get("exp_divisor", 1.0), xf.get("apply_inverse", False))
        ok = ok and np.array_equal(N(outs[0][b]), ref)
    if all(pipeline.ragged_eligible(int(i.shape[0]), int(i.shape[1]), 3) for (im, _) in batches for i in im) and Wo * 3 <= 4096:
        st = pipeline.RaggedMaskChainStream(out_size=(Ho, Wo), **xf)
        got = []
        for (im, a) in batches:
            d = st.push(im, a)
            if d is not None: got.append(d.out)
        got += [d.out for d in st.flush()]
        ok = ok and len(got) == nb and all(bool(torch.equal(g, o)) for g, o in zip(got, outs))
    return ok, (B, nb, Ho, Wo, [tuple(i.shape[:2]) for i in imgs], xf)

for name, gen in (("ragged chain vs per-image + oracle", g_ragged_chain), ("MaskChainStream vs warp_from_masks", g_mask_chain_stream), ("clip_preprocess (+pad) (f3)", g_clip), ("probe_last_query (f4)", g_probe), ("MarginalNet tail (f1)", g_mn_tail), ("MarginalNet forward fused vs stock (f1)", g_marginalnet),
                  ("warp_image_by_attention chain", g_chain_u8), ("attention stack -> warp (bench path)", g_stack_chain),
                  ("attention_axis_maps (A13)", g_att_maps), ("axis_maps_from_pdf (A8-A11)", g_pdf_chain),
                  ("cdf / repair / resample (A9-A10)", g_cdf_stages), ("attn reduce step (A1)", g_attn),
                  ("LANCZOS mask up-sample (A4)", g_lanczos), ("pool24 + gt_marginals (A5-A6)", g_pool_marg),
                  ("revise_mask + safe_softmax (A3,A7)", g_revise_softmax)):
    try:
        run(name, gen)
    except Exception as e:   # noqa: BLE001
        print(f"{name}: EXCEPTION {type(e).__name__}: {e}", flush=True)
