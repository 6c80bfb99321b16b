"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol the header
declares, argument validation works without a GPU, host tables match the oracle, the Python
surface keeps the reference's error behaviour, and there is NO CPU fallback."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import warp_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from attwarp_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    from attwarp_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "attwarp.h")).read()
    tuning_only = set(re.findall(r"(attwarp_\w+)\s*\(", "".join(re.findall(r"#ifdef ATTWARP_TUNING(.*?)#endif", hdr, re.S))))
    declared = set(re.findall(r"ATTWARP_API\s+[\w\s\*]+?\b(attwarp_\w+)\s*\(", hdr)) - tuning_only
    assert len(declared) >= 20 and tuning_only == {"attwarp_debug_set", "attwarp_debug_stream_copy"}
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/attwarp.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.attwarp_version() == 100
    # the product library carries NO test / measurement hook: exactly the declared entry points are exported
    import subprocess
    def exports(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        return {l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("attwarp_")}
    assert exports(_lib.LIB_PATH) == declared
    assert exports(_lib.TUNING_LIB_PATH) == declared | tuning_only


def test_debug_override_swaps_to_the_tuning_library_and_restores(lib):
    """debug_override runs on the tuning flavour, restores the previous value of every key (nesting) and hands calls
    back to the product library afterwards (resetting to -1 clobbered an outer override)."""
    import ctypes
    from attwarp_amd import _lib
    assert _lib.load() is lib
    with _lib.debug_override(remap_rows=5):
        tl = _lib.load()
        assert tl is not lib and hasattr(tl, "attwarp_debug_set")
        with _lib.debug_override(remap_rows=7, remap_variant=1):
            pass
        old = ctypes.c_int(-99)
        assert tl.attwarp_debug_set(b"remap_rows", 5, ctypes.byref(old)) == 0 and old.value == 5      # inner exit restored 5
        assert tl.attwarp_debug_set(b"remap_variant", -1, ctypes.byref(old)) == 0 and old.value == -1
    assert _lib.load() is lib
    old = ctypes.c_int(-99)
    assert _lib.load_tuning().attwarp_debug_set(b"remap_rows", -1, ctypes.byref(old)) == 0 and old.value == -1
    with pytest.raises(_lib.AttWarpError):
        with _lib.debug_override(no_such_key=1):
            pass


def test_c_abi_argument_validation_without_gpu(lib):
    # every entry point validates before it enqueues anything, so these run on a CPU-only box
    assert lib.attwarp_remap_bilinear(None, None, 0, 0, 1, 3, 8, 8, 8, 8, None, None, 0, None) == -1
    assert b"null pointer" in lib.attwarp_last_error()
    buf = (8 * 8 * 3 * 4) * b"\0"
    import ctypes
    p = ctypes.cast(ctypes.create_string_buffer(buf), ctypes.c_void_p)
    assert lib.attwarp_remap_bilinear(p, p, 7, 0, 1, 3, 8, 8, 8, 8, p, p, 0, None) == -1      # bad dtype
    assert lib.attwarp_remap_bilinear(p, p, 0, 0, 1, 3, 8, 8, 8, 8, p, p, 5, None) == -1      # bad mode
    assert lib.attwarp_remap_bilinear(p, p, 0, 0, 1, 9, 8, 8, 8, 8, p, p, 0, None) == -2      # C > 4
    assert lib.attwarp_remap_bilinear(p, p, 0, 0, 0, 3, 8, 8, 8, 8, p, p, 0, None) == -1      # B = 0
    assert lib.attwarp_axis_map_from_cdf(p, 1, 100000, 8, p, None) == -2
    assert lib.attwarp_mask_postproc(p, 1, 24, 4, 10.0, p, None) == -1                         # even kernel
    assert lib.attwarp_attn_reduce_step(p, 3, 1, 1, 1, 8, 0, 0, 0, 1, p, 4, p, None) == -1     # uint8 attention
    assert lib.attwarp_axis_maps_from_attention(p, 0, 1, 8, 8, 8, 8, 9, 1.0, 1.0, 0, p, p, p, None) == -1
    assert lib.attwarp_axis_sums_workspace_bytes(2, 3, 5) == 2 * 8 * 8


def test_stream_step_abi_validation_without_gpu(lib):
    """The two stream-step entry points (attwarp_attn_reduce_and_maps, attwarp_warp_step_fused) validate before they enqueue."""
    import ctypes
    p = ctypes.cast(ctypes.create_string_buffer(1 << 16), ctypes.c_void_p)
    q = ctypes.c_void_p(p.value + 4096)
    am = lib.attwarp_attn_reduce_and_maps
    ok = dict(attn_dtype=0, rows=p, n_rows=2, heads=4, kv_len=640, starts=p, starts_mod=2, ntok=576, steps_out=p, steps_in=q,
              T=1, B=2, g=24, W=64, H=64, W_out=64, H_out=64, inv_x=p, inv_y=p, map_x=p, map_y=p, stream=None)
    def call(**kw):
        a = dict(ok, **kw)
        return am(*[a[k] for k in ok])
    assert call(rows=None) == -1 and b"null pointer" in lib.attwarp_last_error()
    assert call(attn_dtype=3) == -1                      # uint8 attention
    assert call(kv_len=500) == -1                        # ntok > kv_len
    assert call(steps_in=p) == -1 and b"alias" in lib.attwarp_last_error()
    assert call(ntok=572) == -2                          # ntok != g*g
    assert call(g=40, ntok=1600, kv_len=2000) == -2
    assert call(W=9000) == -2
    assert call(B=0) == -1
    fused = lib.attwarp_warp_step_fused
    # (src, dst, layout, B, C, H, W, H_out, W_out, map_x, map_y, mode, attn_dtype, steps_in, T, g, inv_x, inv_y, map_x_next,
    #  map_y_next, rows, n_rows, heads, kv_len, starts, starts_mod, ntok, steps_out, stream)
    base = [p, q, 0, 2, 3, 64, 64, 64, 64, p, p, 1, 0, p, 1, 24, p, p, q, q, p, 2, 4, 640, p, 2, 576, q, None]
    def fcall(**kw):
        names = ["src", "dst", "layout", "B", "C", "H", "W", "H_out", "W_out", "map_x", "map_y", "mode", "attn_dtype", "steps_in",
                 "T", "g", "inv_x", "inv_y", "map_x_next", "map_y_next", "rows", "n_rows", "heads", "kv_len", "starts",
                 "starts_mod", "ntok", "steps_out", "stream"]
        a = dict(zip(names, base)); a.update(kw)
        return fused(*[a[k] for k in names])
    assert fcall(src=None) == -1
    assert fcall(mode=7) == -1
    assert fcall(attn_dtype=3) == -1
    assert fcall(steps_out=p) == -1                      # aliases steps_in
    assert fcall(map_x_next=p) == -1                     # aliases the maps being read
    assert fcall(kv_len=100) == -1
    # steps_out of one call is steps_in [T,B,g*g] of the next: both pieces present => ntok must equal g*g
    assert fcall(ntok=572) == -2 and b"g*g" in lib.attwarp_last_error()
    assert fcall(ntok=572, steps_in=None) != -2 or b"g*g" not in lib.attwarp_last_error()   # A alone: any multiple of 4


def test_step_slots_abi_validation_without_gpu(lib):
    """attwarp_warp_step_fused_slots (two batches per piece and launch) validates before it enqueues."""
    import ctypes
    from attwarp_amd.pipeline import _StepSlot
    buf = ctypes.create_string_buffer(1 << 16)
    base = (ctypes.addressof(buf) + 15) & ~15
    P = [base + 1024 * i for i in range(24)]
    def slots(**kw):
        a = (_StepSlot * 2)()
        a[0] = _StepSlot(P[0], P[1], P[2], P[3], P[4], P[5], P[6], P[7], P[8], P[9])
        a[1] = _StepSlot(P[10], P[11], P[12], P[13], P[14], P[15], P[16], P[17], P[8], P[19])
        for k, v in kw.items():
            setattr(a[1], k, v)
        return a
    f = lib.attwarp_warp_step_fused_slots
    def call(a, nslots=2, **kw):
        args = dict(layout=0, B=2, C=3, H=64, W=64, H_out=64, W_out=64, mode=1, attn_dtype=0, T=1, g=24, inv_x=P[20], inv_y=P[21],
                    n_rows=2, heads=4, kv_len=640, starts_mod=2, ntok=576)
        args.update(kw)
        return f(ctypes.cast(a, ctypes.c_void_p) if a is not None else None, nslots, *[args[k] for k in
                 ("layout", "B", "C", "H", "W", "H_out", "W_out", "mode", "attn_dtype", "T", "g", "inv_x", "inv_y", "n_rows", "heads",
                  "kv_len", "starts_mod", "ntok")], None)
    assert call(None) == -1 and b"slot table" in lib.attwarp_last_error()
    assert call(slots(), nslots=3) == -1
    assert call(slots(src=None)) == -1
    assert call(slots(dst=P[1])) == -1 and b"different buffers" in lib.attwarp_last_error()
    assert call(slots(src=P[10] + 4)) == -1 and b"alignment" in lib.attwarp_last_error()
    assert call(slots(rows=None)) == -1 and b"same pieces" in lib.attwarp_last_error()
    assert call(slots(map_x_next=P[2])) == -1 and b"alias" in lib.attwarp_last_error()      # slot 1 writes the maps slot 0 reads
    assert call(slots(steps_out=P[4])) == -1                                                  # ... the step maps slot 0's M reads
    assert call(slots(), ntok=572) == -2
    assert call(slots(), mode=9) == -1


def test_mask_chain_step_abi_validation_without_gpu(lib):
    """attwarp_mask_chain_step (the one-launch step of the main_batched chain) validates before it enqueues."""
    import ctypes
    buf = ctypes.create_string_buffer(1 << 16)
    base = (ctypes.addressof(buf) + 15) & ~15
    P = [ctypes.c_void_p(base + 1024 * i) for i in range(16)]
    names = ["images", "out", "B", "C", "H", "W", "H_out", "W_out", "map_x", "map_y", "sums_in", "map_x_next", "map_y_next",
             "mota_in", "sums_out", "rev_in", "bounds_x", "kk_x", "ksize_x", "bounds_y", "kk_y", "ksize_y", "mota_out", "masks",
             "g", "kernel_size", "enhance_coe", "rev_out", "transform", "exp_scale", "exp_divisor", "apply_inverse", "transform_lut",
             "stream"]
    ok = dict(images=P[0], out=P[1], B=2, C=3, H=64, W=64, H_out=80, W_out=80, map_x=P[2], map_y=P[3], sums_in=P[4],
              map_x_next=P[5], map_y_next=P[6], mota_in=P[7], sums_out=P[8], rev_in=P[9], bounds_x=P[10], kk_x=P[11], ksize_x=8,
              bounds_y=P[12], kk_y=P[13], ksize_y=8, mota_out=P[14], masks=P[15], g=24, kernel_size=3, enhance_coe=10.0,
              rev_out=ctypes.c_void_p(base + 1024 * 16), transform=0, exp_scale=1.0, exp_divisor=1.0, apply_inverse=0,
              transform_lut=None, stream=None)
    f = lib.attwarp_mask_chain_step
    def call(**kw):
        a = dict(ok, **kw)
        return f(*[a[k] for k in names])
    assert call(images=None) == -1 and b"null" in lib.attwarp_last_error()
    assert call(masks=None) == -1
    assert call(B=0) == -1
    assert call(kernel_size=4) == -1 and b"odd" in lib.attwarp_last_error()
    assert call(transform=5) == -1 and b"unknown transform" in lib.attwarp_last_error()
    assert call(transform=2) == -1 and b"transform_lut" in lib.attwarp_last_error()      # sqrt / exp / log need the table
    assert call(transform=3) == -1 and call(transform=4) == -1
    assert lib.attwarp_attention_transform_lut(2, 1.0, 1.0, None, None) == -1
    assert lib.attwarp_attention_transform_lut(7, 1.0, 1.0, P[0], None) == -1
    assert call(map_x_next=P[2]) == -1 and b"alias" in lib.attwarp_last_error()
    assert call(sums_out=P[4]) == -1 and call(mota_out=P[7]) == -1 and call(rev_out=P[9]) == -1
    assert call(g=40) == -2
    assert call(W=66) == -2                              # W % 4 != 0: not the column-strip up-sampler
    assert call(W=24) == -2                              # no horizontal up-sampling
    assert call(ksize_y=7) == -2
    assert call(W=2048, H=8) == -2                       # rows wider than 4096 bytes: not the integer resample


def test_probe_abi_validation_without_gpu(lib):
    import ctypes
    p = ctypes.cast(ctypes.create_string_buffer(4096), ctypes.c_void_p)
    a = ctypes.c_void_p((p.value + 15) & ~15)
    f = lib.attwarp_attn_probe_last_query
    assert f(a, a, 1, 2, 4, 2, 64, 48, 256, 64, 6144, 3072, 64, None, a, 24, 0.125, a, None, None) == -1   # ws null
    assert f(a, a, 1, 2, 4, 3, 64, 48, 256, 64, 6144, 3072, 64, None, a, 24, 0.125, a, a, None) == -1      # heads % kv_heads
    assert f(a, a, 3, 2, 4, 2, 64, 48, 256, 64, 6144, 3072, 64, None, a, 24, 0.125, a, a, None) == -1      # uint8
    assert f(a, a, 1, 2, 4, 2, 60, 48, 240, 60, 5760, 2880, 60, None, a, 24, 0.125, a, a, None) == -2      # D % 8
    assert f(a, a, 1, 2, 4, 2, 64, 48, 256, 64, 6144, 3072, 65, None, a, 24, 0.125, a, a, None) == -2      # stride
    assert b"16 bytes" in lib.attwarp_last_error()
    assert f(a, a, 1, 2, 4, 2, 64, 20000, 256, 64, 6144, 3072, 64, None, a, 24, 0.125, a, a, None) == -2   # kv too long
    assert f(a, a, 1, 2, 4, 2, 64, 16, 256, 64, 6144, 3072, 64, None, a, 24, 0.125, a, a, None) == -1      # ntok > kv
    assert lib.attwarp_attn_probe_workspace_bytes(1, 2, 4, 24) == 16 + 2 * 4 * 24 * 2


def test_register_probe_plumbing_on_hf_llama():
    """register_probe wires the target layer only, sees post-RoPE (query, key), the model's own mask and
    scaling, leaves the model's output untouched and restores everything on removal (no GPU needed: the
    probe itself is replaced by a recorder)."""
    transformers = pytest.importorskip("transformers")
    from attwarp_amd import attention_extraction as ae
    cfg = transformers.LlamaConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4,
                                   num_key_value_heads=2, vocab_size=100, head_dim=16)
    cfg._attn_implementation = "sdpa"
    torch.manual_seed(0)
    m = transformers.LlamaModel(cfg).eval()
    holder = type("Holder", (), {})()
    holder.model = m
    hl = ae.BatchMaskHookLogger(holder, "cpu", layer_index=1)
    calls = []
    hl._probe_attention = lambda q, k, mask, scaling=None: calls.append((tuple(q.shape), tuple(k.shape), mask, scaling))
    hl.set_batch_image_token_ranges([2, 3], [10, 11])
    ids = torch.randint(0, 100, (2, 12))
    am = torch.ones(2, 12, dtype=torch.long)
    am[1, :3] = 0
    with torch.no_grad():
        ref = m(input_ids=ids, attention_mask=am).last_hidden_state
        hl.register_probe()
        out = m(input_ids=ids, attention_mask=am).last_hidden_state
    assert torch.equal(ref, out)
    assert len(calls) == 1                                           # one layer, one call
    qs, ks, mask, scaling = calls[0]
    assert qs == (2, 4, 12, 16) and ks == (2, 2, 12, 16) and scaling == 0.25
    assert mask.shape == (2, 1, 12, 12) and not bool(mask[1, 0, -1, :3].any()) and bool(mask[1, 0, -1, 3:].all())
    hl.remove_hook_and_unpatch()
    assert m.layers[1].self_attn.config is cfg
    calls.clear()
    with torch.no_grad():
        m(input_ids=ids, attention_mask=am)
    assert calls == []
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ae.probe_last_query(torch.rand(2, 4, 16), torch.rand(2, 2, 12, 16), torch.zeros(2, dtype=torch.int32), 8)


def test_marginalnet_tail_abi_validation_without_gpu(lib):
    import ctypes
    p = ctypes.cast(ctypes.create_string_buffer(256), ctypes.c_void_p)
    assert lib.attwarp_masked_token_mean(p, 3, p, 1, 4, 8, p, None) == -1          # uint8 tokens
    assert lib.attwarp_masked_token_mean(p, 0, None, 1, 4, 8, p, None) == -1       # null mask
    assert lib.attwarp_masked_token_mean(p, 0, p, 1, 0, 8, p, None) == -1          # Lt = 0
    assert lib.attwarp_film_axis_means(p, p, 1, 2, 64, 64, p, p, None) == -2       # tile too large
    assert lib.attwarp_film_axis_means(p, None, 1, 2, 4, 4, p, p, None) == -1
    from attwarp_amd import model
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model.masked_token_mean(torch.rand(2, 3, 8), torch.ones(2, 3, 1))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model.film_axis_means(torch.rand(2, 4, 6, 6), torch.rand(2, 8))


def test_batch_image_token_ranges_bookkeeping():
    """reference functions.py:276-291: starts shift by the left padding of the expanded prompts."""
    from attwarp_amd import attention_extraction as ae
    starts, ends, pads = ae.batch_image_token_ranges([40, 47, 43], [35, 36, 35])
    assert pads == [7, 0, 4]                                 # expanded lengths 615, 622, 618
    assert starts == [42, 36, 39] and ends == [618, 612, 615]
    assert ae.batch_image_token_ranges([], []) == ([], [], [])
    with pytest.raises(ValueError):
        ae.batch_image_token_ranges([1, 2], [0])
    hl = ae.BatchMaskHookLogger(None, "cpu")
    hl.set_batch_image_token_ranges(starts, ends)
    assert hl.batch_size == 3 and hl.image_token_starts == starts


def test_no_cpu_fallback():
    from attwarp_amd import checkpoint_utils as cu, model, attention_extraction as ae
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        cu.cdf_from_density(torch.rand(2, 16))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        cu.warp_from_cdf_torch(torch.rand(1, 3, 8, 8), torch.rand(1, 8), torch.rand(1, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model.safe_softmax(torch.rand(2, 24))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ae.revise_mask(torch.rand(24, 24))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: no product file may import, include, link or load it
    (comments that merely cite it are fine)."""
    pkg = os.path.join(ROOT, "attwarp_amd")
    bad = re.compile(r"^\s*(from\s+oracle|import\s+oracle)|#\s*include\s*[\"<][^\n]*oracle|liboracle|c_oracle|warp_oracle\s*import",
                     re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                src = open(os.path.join(dirpath, f)).read()
                m = bad.search(src)
                assert m is None, f"{f} uses the oracle: {m.group(0)!r}"


def test_reference_error_behaviour():
    from attwarp_amd import checkpoint_utils as cu
    with pytest.raises(AssertionError):
        cu.warp_from_cdf_torch(torch.rand(3, 8, 8), torch.rand(1, 8), torch.rand(1, 8))
    with pytest.raises(ValueError, match="image width"):
        cu.warp_from_cdf_torch(torch.rand(1, 3, 8, 8), torch.rand(1, 7), torch.rand(1, 8))
    with pytest.raises(ValueError, match="image height"):
        cu.warp_from_cdf_torch(torch.rand(1, 3, 8, 8), torch.rand(1, 8), torch.rand(1, 9))
    with pytest.raises(ValueError, match="1D/2D/3D"):
        cu.upsample_pdf_right_inverse(torch.rand(1, 1, 1, 24), 336)


def test_set_transform_function_semantics(capsys):
    from attwarp_amd import new_method as nm
    assert nm.set_transform_function("sqrt", 2.0, 3.0, True) == "sqrt"
    assert (nm.ATTENTION_TRANSFORM, nm.EXP_SCALE, nm.EXP_DIVISOR, nm.APPLY_INVERSE_TO_MARGINALS) == ("sqrt", 2.0, 3.0, True)
    assert nm.set_transform_function("bogus") == "identity"
    assert "Unknown transform" in capsys.readouterr().out
    assert nm.ATTENTION_TRANSFORM == "identity" and nm.APPLY_INVERSE_TO_MARGINALS is False


def test_save_warped_image_swallows_errors(tmp_path, capsys):
    from attwarp_amd import new_method as nm
    ok = nm.save_warped_image(str(tmp_path / "missing.png"), np.ones((4, 4)), None, None, str(tmp_path / "o.png"))
    assert ok is False
    ok = nm.save_warped_image(np.zeros((4, 4, 3), np.uint8), np.ones((2, 2, 2, 2)), None, None, str(tmp_path / "o.png"))
    assert ok is False          # 4-D attention map -> ValueError inside, swallowed


def test_att_map_coercion():
    from attwarp_amd import new_method as nm
    from PIL import Image
    a = nm._coerce_att_map([], 7, 5)
    assert a.shape == (5, 7) and np.all(a == 128)
    a = nm._coerce_att_map([np.ones((3, 4))], 7, 5)
    assert a.shape == (3, 4)
    a = nm._coerce_att_map(Image.fromarray(np.zeros((6, 8), np.uint8), mode="L"), 1, 1)
    assert a.shape == (6, 8)
    a = nm._coerce_att_map(np.ones((3, 4, 3)) * np.array([1, 2, 3]), 1, 1)
    assert a.shape == (3, 4) and np.allclose(a, 2)


@pytest.mark.parametrize("L", [336, 500, 512, 1024, 37])
def test_right_inverse_table_matches_oracle(L):
    from attwarp_amd import _tables
    assert np.array_equal(_tables._right_inverse_inv_host(24, L, 1e-8), O.right_inverse_core(24, L, 1e-8))


@pytest.mark.parametrize("io", [(24, 336), (24, 1024), (24, 500), (24, 17), (24, 48)])
def test_lanczos_tables_match_oracle(io):
    from attwarp_amd import _tables
    b, k, ks = _tables._lanczos_tables_host(*io)
    bo, ko, kso = O.pil_lanczos_coeffs(*io)
    assert ks == kso and np.array_equal(b, bo) and np.array_equal(k, ko)


@pytest.mark.parametrize("io", [(500, 336), (375, 336), (64, 448), (40, 17)])
def test_bicubic_tables_match_oracle(io):
    from attwarp_amd import _tables
    b, k, ks = _tables._lanczos_tables_host(io[0], io[1], "bicubic")
    bo, ko, kso = O.pil_resample_coeffs(io[0], io[1], "bicubic")
    assert ks == kso and np.array_equal(b, bo) and np.array_equal(k, ko)
    bi, ki, ksi = _tables._lanczos_tables_host(336, 336, "bicubic")          # identity pass
    assert ksi == 1 and np.all(ki == 1 << 22) and np.array_equal(bi[:, 0], np.arange(336))


def test_marginalnet_logits_match_reference_golden(golden):
    """MarginalNet on stock torch ops with the reference's parameter names: load the seeded
    reference state_dict, compare softmax(logits) with the captured (px, py)."""
    from attwarp_amd import model
    g = golden("marginalnet")
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd|")}
    net = model.MarginalNet(32, 48, hidden=16).eval()
    model.load_reference_checkpoint(net, {"epoch": 1, "model": sd})
    with torch.no_grad():
        lx, ly = net.forward_logits(torch.from_numpy(g["fmap"]), 24, 24, torch.from_numpy(g["ttok"]),
                                    torch.from_numpy(g["tmask"]))
    np.testing.assert_allclose(O.safe_softmax(lx.numpy()), g["px"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(O.safe_softmax(ly.numpy()), g["py"], rtol=2e-5, atol=1e-7)
    full = model.MarginalNet(1024, 4096, 256)
    assert sum(p.numel() for p in full.parameters()) == 2755074     # the 11.0 MB payload


def test_shard_range_partitions():
    from attwarp_amd.dist import shard_range
    for n in (0, 1, 7, 64, 2048, 2051):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_reference_checkpoint_file_roundtrip(tmp_path):
    """load_reference_checkpoint on a real file in the trainer's format (MN/trainer.py:660-683:
    ``{"epoch", "model", "opt", "cfg"}``), on a bare state_dict file and on an in-memory dict; a checkpoint of the wrong
    architecture fails loudly.  (CPU: only the loader, no compute.)"""
    import torch
    from attwarp_amd import model
    torch.manual_seed(5)
    src = model.MarginalNet(12, 20, hidden=8)
    opt = torch.optim.AdamW(src.parameters(), lr=3e-4, weight_decay=1e-4)
    full = tmp_path / "marginal_net_epoch_3.pt"
    torch.save({"epoch": 3, "model": src.state_dict(), "opt": opt.state_dict(), "cfg": {"hidden": 8, "seed": 13}}, full)
    bare = tmp_path / "bare.pt"
    torch.save(src.state_dict(), bare)
    for source in (str(full), str(bare), {"model": src.state_dict()}, src.state_dict()):
        dst = model.MarginalNet(12, 20, hidden=8)
        model.load_reference_checkpoint(dst, source)
        for (ka, va), (kb, vb) in zip(src.state_dict().items(), dst.state_dict().items()):
            assert ka == kb and torch.equal(va, vb)
    with pytest.raises(RuntimeError):
        model.load_reference_checkpoint(model.MarginalNet(12, 20, hidden=16), str(full))


def test_golden_marginalnet_full_recipe():
    """The seeded weight recipe the GPU test regenerates must be the one the golden was made with."""
    import numpy as np
    from conftest import marginalnet_full_state
    from attwarp_amd import model
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "marginalnet_full.npz"))
    net = model.MarginalNet(1024, 4096, hidden=256)
    sd = marginalnet_full_state({k: tuple(v.shape) for k, v in net.state_dict().items()})
    assert abs(sum(v.double().sum().item() for v in sd.values()) - float(g["sd_checksum"])) < 1e-6
    assert sum(v.numel() for v in sd.values()) == int(g["n_params"])


def test_bench_lists_and_validates_legs():
    """bench.py --list-legs names every secondary leg; an unknown leg is refused before anything touches a GPU."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--list-legs"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0
    names = [l.split()[0] for l in r.stdout.splitlines() if l.strip()]
    assert {"exact", "chw", "fused", "distributions", "336", "fp16_attention", "main_batched", "pool_input", "config5"} <= set(names)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--legs", "no_such_leg"], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode != 0 and "unknown leg" in (r.stderr + r.stdout)


def test_pair_slots_views():
    """pipeline.pair_slots (two batches of a stream as one): views, not copies; adjacency and parity are checked."""
    from attwarp_amd import pipeline
    r = torch.arange(4 * 3 * 2, dtype=torch.float32).reshape(4, 3, 2)
    ps = pipeline.pair_slots(list(r))
    assert len(ps) == 2 and tuple(ps[0].shape) == (6, 2) and torch.equal(ps[1], r[2:4].reshape(6, 2))
    ps[0][4, 1] = -1.0
    assert float(r[1, 1, 1]) == -1.0                               # a view of the ring's memory
    with pytest.raises(ValueError):
        pipeline.pair_slots([torch.zeros(3, 2), torch.zeros(3, 2)])
    with pytest.raises(ValueError):
        pipeline.pair_slots(list(r[:3]))



# ---- the ragged main_batched chain: host side (coefficient helper, batch table, argument validation) ------------------------
def test_pil_coeffs_helper_matches_oracle_over_many_sizes(lib):
    """attwarp_pil_coeffs_8bpc (the library's host helper behind _tables.lanczos_tables) against the oracle's restatement of
    Pillow's precompute_coeffs / normalize_coeffs_8bpc for the sizes a ragged batch brings (every 7th size up to 1400 and the
    TextVQA-like ones), zero padding to 8 columns included."""
    from attwarp_amd import _tables
    sizes = sorted(set(range(25, 1400, 7)) | {333, 375, 427, 500, 501, 640, 683, 768, 1023, 1024, 1365, 2048, 13})
    for n in sizes:
        b, k, ks = _tables._lanczos_tables_host(24, n)
        bo, ko, kso = O.pil_lanczos_coeffs(24, n)
        assert ks == kso and np.array_equal(b, bo) and np.array_equal(k, ko), n
    bounds = np.zeros((683, 2), np.int32); kk = np.full((683, 8), -1, np.int32)
    assert lib.attwarp_pil_coeffs_8bpc(24, 683, 0, bounds.ctypes.data, kk.ctypes.data, 8) == 7
    bo, ko, _ = O.pil_lanczos_coeffs(24, 683)
    assert np.array_equal(kk[:, :7], ko) and (kk[:, 7] == 0).all() and np.array_equal(bounds, bo)
    assert lib.attwarp_pil_coeffs_8bpc(24, 683, 0, bounds.ctypes.data, kk.ctypes.data, 6) == -1      # 7 taps do not fit
    assert lib.attwarp_pil_coeffs_8bpc(24, 683, 5, bounds.ctypes.data, kk.ctypes.data, 8) == -1      # unknown filter
    assert lib.attwarp_pil_coeffs_8bpc(24, 683, 0, None, kk.ctypes.data, 8) == -1


def _ragged_records(sizes, base=0x10000):
    from attwarp_amd import _lib
    rec = (_lib.RaggedImage * len(sizes))()
    for i, (h, w) in enumerate(sizes):
        rec[i].image = base + 0x1000000 * i
        rec[i].bounds_x = rec[i].kk_x = rec[i].bounds_y = rec[i].kk_y = base
        rec[i].H, rec[i].W, rec[i].ksize_x = h, w, 8
    return rec


def test_ragged_plan_table_layout_without_gpu(lib):
    """attwarp_ragged_plan writes a position-independent table: header, per-image records, numpy's pairwise plans of the
    distinct sizes, and the block maps of the up-sampling / marginals stages."""
    import ctypes
    from attwarp_amd import _lib, pipeline
    sizes = [(768, 1024), (1024, 683), (1024, 1024), (375, 500), (500, 333), (427, 640), (1024, 683)]
    rec = _ragged_records(sizes)
    B = len(sizes)
    n = lib.attwarp_ragged_table_bytes(ctypes.byref(rec), B, 3, 24, 500, 500)
    assert n > 0 and n % 16 == 0
    buf = np.zeros(n + 64, np.uint8)
    assert lib.attwarp_ragged_plan(ctypes.byref(rec), B, 3, 24, 500, 500, buf.ctypes.data, n) == 0
    assert not buf[n:].any()                                       # nothing written behind the table
    h = _lib.RaggedHeader.from_address(buf.ctypes.data)
    assert (h.B, h.C, h.g, h.H_out, h.W_out, h.table_bytes) == (B, 3, 24, 500, 500, n)
    assert h.rows_per_block == 16 and h.blocks_per_image == 32 and h.nR == 32 * B and h.kd == 2 and h.max_hw == 1024
    img = (pipeline._RaggedImageDev * B).from_address(buf.ctypes.data + h.off_images)
    plans = {}
    # plans: same leaves as numpy's recursion (test_oracle_golden pins that recursion against np.sum itself)
    def leaves(off, m, out):
        if m <= 128:
            out.append((off, m)); return
        n2 = m // 2; n2 -= n2 % 8
        leaves(off, n2, out); leaves(off + n2, m - n2, out)
    plan_words = 4 + 64 + 64 + 32
    praw = np.frombuffer(buf, np.int32, h.nplans * plan_words, h.off_plans).reshape(h.nplans, plan_words)
    distinct = sorted({s for hw in sizes for s in hw})
    assert h.nplans == len(distinct)
    mota_end = sums_end = 0
    nL = nP = 0
    for b, (H, W) in enumerate(sizes):
        r = img[b]
        assert (r.H, r.W, r.image) == (H, W, rec[b].image)
        pw, ph = praw[r.plan_w], praw[r.plan_h]
        assert pw[3] == W and ph[3] == H
        for p, m in ((pw, W), (ph, H)):
            want = []; leaves(0, m, want)
            assert p[0] == len(want) and [(int(p[4 + i]), int(p[68 + i])) for i in range(p[0])] == want
        assert r.ki == -(-(-(-W * 3 // 4)) // 256)
        assert r.mota_off >= mota_end and r.mota_off % 256 == 0
        mota_end = r.mota_off + H * W
        assert r.sums_off == sums_end
        sums_end += W + H * int(pw[0])
        assert sums_end * 8 - r.sums_off * 8 == lib.attwarp_axis_sums_workspace_bytes(1, H, W)
        nstrips = -(-W // 256)
        assert r.l_nchunks * r.l_rows_per_chunk >= H > (r.l_nchunks - 1) * r.l_rows_per_chunk
        nL += nstrips * r.l_nchunks; nP += int(pw[0])
    assert h.mota_bytes >= mota_end and h.sums_bytes == sums_end * 8 and (h.nL, h.nP) == (nL, nP)
    lmap = np.frombuffer(buf, np.uint32, h.nL, h.off_lmap); pmap = np.frombuffer(buf, np.uint32, h.nP, h.off_pmap)
    # the stages visit the images largest first (stable): the longest blocks of a launch start first
    order = np.frombuffer(buf, np.uint32, B, h.off_order).tolist()
    assert order == sorted(range(B), key=lambda b: -sizes[b][0] * sizes[b][1]) and order[:3] == [2, 0, 1]
    for m, per in ((lmap, [-(-w // 256) * img[b].l_nchunks for b, (_, w) in enumerate(sizes)]),
                   (pmap, [int(praw[img[b].plan_w][0]) for b in range(B)])):
        want = [(b, s) for b in order for s in range(per[b])]
        assert [(int(e & 0xffff), int(e >> 16)) for e in m] == want
    # limits
    def rc(sz, C=3, Ho=500, Wo=500, g=24):
        r = _ragged_records(sz)
        return lib.attwarp_ragged_plan(ctypes.byref(r), len(sz), C, g, Ho, Wo, buf.ctypes.data, buf.size)
    assert rc([(100, 1366)]) == -2 and b"4096" in lib.attwarp_last_error()       # rows of 4098 bytes
    assert rc([(100, 1365)]) == 0
    assert rc([(24, 100)]) == -2 and rc([(100, 24)]) == -2                        # no up-sampling on an axis
    assert rc([(9000, 100)]) == -2
    assert rc([(100, 100)], Wo=1400) == -2
    assert rc([(100, 100)], C=5) == -2 and rc([(100, 100)], g=40) == -2
    r = _ragged_records([(100, 100)]); r[0].kk_y = None
    assert lib.attwarp_ragged_plan(ctypes.byref(r), 1, 3, 24, 500, 500, buf.ctypes.data, buf.size) == -1
    assert lib.attwarp_ragged_plan(ctypes.byref(rec), B, 3, 24, 500, 500, buf.ctypes.data, n - 16) == -1
    assert lib.attwarp_ragged_table_bytes(ctypes.byref(_ragged_records([(100, 1366)])), 1, 3, 24, 500, 500) == 0
    assert pipeline.ragged_eligible(100, 1365, 3) and not pipeline.ragged_eligible(100, 1366, 3)
    assert not pipeline.ragged_eligible(24, 100, 3) and not pipeline.ragged_eligible(100, 9000, 1)


def test_mask_chain_ragged_abi_validation_without_gpu(lib):
    """attwarp_mask_chain_ragged validates before it enqueues (no GPU here: every call must fail in validation)."""
    import ctypes
    rec = _ragged_records([(64, 100), (90, 77)])
    n = lib.attwarp_ragged_table_bytes(ctypes.byref(rec), 2, 3, 24, 80, 88)
    t1 = np.zeros(n, np.uint8); t2 = np.zeros(n, np.uint8)
    assert lib.attwarp_ragged_plan(ctypes.byref(rec), 2, 3, 24, 80, 88, t1.ctypes.data, n) == 0
    assert lib.attwarp_ragged_plan(ctypes.byref(rec), 2, 3, 24, 96, 88, t2.ctypes.data, n) == 0      # another output size
    P = [ctypes.c_void_p(0x100000 + 4096 * i) for i in range(16)]
    names = ["r_host", "r_dev", "out", "map_x", "map_y", "f_host", "f_dev", "sums_in", "map_x_next", "map_y_next", "p_host", "p_dev",
             "mota_in", "sums_out", "l_host", "l_dev", "rev_in", "mota_out", "masks", "B_masks", "g", "kernel_size", "enhance_coe",
             "rev_out", "transform", "exp_scale", "exp_divisor", "apply_inverse", "transform_lut", "stream"]
    T = ctypes.c_void_p(t1.ctypes.data)
    ok = dict(r_host=T, r_dev=P[0], out=P[1], map_x=P[2], map_y=P[3], f_host=T, f_dev=P[0], sums_in=P[4], map_x_next=P[5],
              map_y_next=P[6], p_host=T, p_dev=P[0], mota_in=P[7], sums_out=P[8], l_host=T, l_dev=P[0], rev_in=P[9], mota_out=P[10],
              masks=P[11], B_masks=2, g=24, kernel_size=3, enhance_coe=10.0, rev_out=P[12], transform=0, exp_scale=1.0,
              exp_divisor=1.0, apply_inverse=0, transform_lut=None, stream=None)
    f = lib.attwarp_mask_chain_ragged
    def call(**kw):
        a = dict(ok, **kw)
        return f(*[a[k] for k in names])
    none = dict(r_host=None, f_host=None, p_host=None, l_host=None, masks=None)
    assert call(**none) == -1 and b"no stage" in lib.attwarp_last_error()
    assert call(r_dev=None) == -1 and call(out=None) == -1 and call(sums_in=None) == -1 and call(mota_in=None) == -1
    assert call(rev_in=None) == -1 and call(rev_out=None) == -1
    assert call(map_x_next=P[2]) == -1 and b"alias" in lib.attwarp_last_error()
    assert call(sums_out=P[4]) == -1 and call(mota_out=P[7]) == -1 and call(rev_out=P[9]) == -1
    assert call(kernel_size=4) == -1 and call(g=16) == -1 and call(B_masks=0) == -1
    assert call(transform=9) == -1 and b"unknown transform" in lib.attwarp_last_error()
    assert call(transform=4) == -1 and b"transform_lut" in lib.attwarp_last_error()
    assert call(f_host=ctypes.c_void_p(t2.ctypes.data)) == -1 and b"share" in lib.attwarp_last_error()
    junk = np.zeros(n, np.uint8)
    assert call(p_host=ctypes.c_void_p(junk.ctypes.data)) == -1 and b"attwarp_ragged_plan" in lib.attwarp_last_error()


def test_ragged_stream_schedule_without_gpu(monkeypatch):
    """RaggedMaskChainStream's schedule is host logic: for any number of batches every batch gets V, L, P, F, R exactly once,
    in that order, one stage per launch, and a launch never runs two stages of one batch (they depend on each other);
    push returns a batch right after the launch that ran its R; the ring form primes / drains the same way."""
    from attwarp_amd import pipeline
    log = []                                   # one dict per launch: stage -> batch id

    class FakeBatch:
        def __init__(self, images, out_size, g, out=None):
            self.id = images
            self._dev = None
            self.masks = None

    def fake_launch(R=None, F=None, P=None, L=None, V=None, enhance_coe=10, kernel_size=3, **transform_kw):
        assert set(transform_kw) == {"transform", "exp_scale", "exp_divisor", "apply_inverse"}
        log.append({k: b.id for k, b in (("R", R), ("F", F), ("P", P), ("L", L), ("V", V)) if b is not None})

    monkeypatch.setattr(pipeline, "RaggedBatch", FakeBatch)
    monkeypatch.setattr(pipeline, "ragged_chain_launch", fake_launch)

    class FakeMask:
        def float(self): return self
        def contiguous(self): return self

    for n in range(1, 12):
        log.clear()
        st = pipeline.RaggedMaskChainStream(out_size=(8, 8))
        done = []
        for i in range(n):
            d = st.push(i, FakeMask())
            if d is not None:
                assert log[-1].get("R") == d.id            # returned right behind the launch that resampled it
                done.append(d.id)
        done += [d.id for d in st.flush()]
        assert done == list(range(n))
        per_batch = {i: [] for i in range(n)}
        for launch in log:
            assert len(set(launch.values())) == len(launch)   # five DIFFERENT batches per launch
            for stage, b in launch.items():
                per_batch[b].append(stage)
        assert all(v == list("VLPFR") for v in per_batch.values()), per_batch
        assert len(log) == n + 4
    # ring form: prime -> steps -> drain leave every slot with all five stages an equal number of times
    for nring, steps in ((5, 5), (6, 12), (7, 3)):
        log.clear()
        st = pipeline.RaggedMaskChainStream(out_size=(8, 8))
        st.ring([FakeBatch(i, None, None) for i in range(nring)])
        st.prime(); st.run(steps); st.drain_ring()
        count = {}
        for launch in log:
            for stage, b in launch.items():
                count[(b, stage)] = count.get((b, stage), 0) + 1
        total = steps + 4                                     # batches that went through the pipeline
        for b in range(nring):
            per = [count.get((b, s), 0) for s in "VLPFR"]
            assert len(set(per)) == 1 and per[0] == (total // nring + (1 if b < total % nring else 0)), (nring, steps, b, per)
