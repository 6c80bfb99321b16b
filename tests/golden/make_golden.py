#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE.

Run in the build container only (``/root/reference`` does not exist on the GPU
box):  ``python tests/golden/make_golden.py``

The reference's Python files are loaded by path with small stub modules for the
third-party packages that are absent here (cv2, torchvision, llava, ...).  The
``cv2.remap`` stub records the float32 maps the reference hands to OpenCV; the
resample itself is not available (OpenCV is neither vendored nor installed), so
no golden exists for it -- "parity unpinned" at that boundary.  ``--with-opencv`` writes that
golden (remap_cv2.npz) on a machine where the real OpenCV imports.

Only inputs and the reference's outputs are stored (data, never source).
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch
from PIL import Image

REF = os.environ.get("ATTWARP_REFERENCE", "/root/reference")
AGW = os.path.join(REF, "Attention Guided Warping")
MN = os.path.join(REF, "model", "marginalnet_full_dataset")
OUT = os.path.dirname(os.path.abspath(__file__))

CAPTURED = {}


def _install_stubs():
    cv2 = types.ModuleType("cv2")
    cv2.INTER_LINEAR = 1
    cv2.INTER_NEAREST = 0
    cv2.BORDER_REPLICATE = 1
    cv2.COLOR_RGB2BGR = 4
    cv2.COLOR_BGR2RGB = 4
    cv2.COLOR_GRAY2BGR = 8
    cv2.COLOR_BGR2GRAY = 6
    cv2.COLORMAP_JET = 2
    cv2.NORM_MINMAX = 32

    def remap(img, map_x, map_y, interpolation=None, borderMode=None):
        CAPTURED["map_x"] = np.array(map_x)
        CAPTURED["map_y"] = np.array(map_y)
        shape = map_x.shape + (img.shape[2:] if img.ndim == 3 else ())
        return np.zeros(shape, dtype=img.dtype)

    def cvtColor(img, code):
        return img[..., ::-1].copy() if img.ndim == 3 and img.shape[2] == 3 else img

    cv2.remap = remap
    cv2.cvtColor = cvtColor
    # blend_mask's overlay tail (llava.py:255-270): visualisation only, shape-correct stand-ins
    cv2.normalize = lambda src, dst, a, b, norm_type: src
    cv2.applyColorMap = lambda m, cmap: np.zeros(m.shape + (3,), np.uint8)
    cv2.addWeighted = lambda a, wa, b, wb, g: a
    cv2.imwrite = lambda *a, **k: True
    cv2.imread = lambda *a, **k: None
    cv2.resize = lambda img, size, interpolation=None: img
    sys.modules["cv2"] = cv2

    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")

    class ToPILImage:
        def __call__(self, pic):
            if pic.is_floating_point():
                pic = pic.mul(255).byte()
            return Image.fromarray(pic.squeeze(0).numpy(), mode="L")

    tvt.ToPILImage = ToPILImage
    tv.transforms = tvt
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tvt

    for name in ["llava", "llava.constants", "llava.conversation", "llava.model", "llava.model.builder",
                 "llava.utils", "llava.mm_utils", "requests"]:
        m = types.ModuleType(name)
        sys.modules[name] = m
    sys.modules["llava.constants"].IMAGE_TOKEN_INDEX = -200
    sys.modules["llava.constants"].DEFAULT_IMAGE_TOKEN = "<image>"
    sys.modules["llava.constants"].DEFAULT_IM_START_TOKEN = "<im_start>"
    sys.modules["llava.constants"].DEFAULT_IM_END_TOKEN = "<im_end>"
    sys.modules["llava.constants"].IMAGE_PLACEHOLDER = "<image-placeholder>"
    sys.modules["llava.conversation"].conv_templates = {}
    sys.modules["llava.conversation"].SeparatorStyle = type("SeparatorStyle", (), {"TWO": 2})
    sys.modules["llava.model.builder"].load_pretrained_model = lambda *a, **k: None
    sys.modules["llava.utils"].disable_torch_init = lambda: None
    for fn in ["process_images", "tokenizer_image_token", "get_model_name_from_path", "KeywordsStoppingCriteria"]:
        setattr(sys.modules["llava.mm_utils"], fn, lambda *a, **k: None)
    # transformers 5.x dropped MaxNewTokensCriteria (reference pins 4.37.2): stub the module
    m = types.ModuleType("transformers.generation.stopping_criteria")
    m.StoppingCriteria = object
    m.MaxNewTokensCriteria = object
    sys.modules["transformers.generation.stopping_criteria"] = m


def pool_input(S: int) -> np.ndarray:
    """Deterministic [2,1,S,S] float32 attention used for the pooling goldens
    (tests import this recipe instead of storing megabytes)."""
    rng = np.random.default_rng(1300 + S)
    A = rng.random((2, 1, S, S), dtype=np.float32)
    A[0, 0, S // 3: S // 3 + S // 8, S // 2: S // 2 + S // 8] += np.float32(5.0)
    return A


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def clip_input(name: str) -> np.ndarray:
    """Deterministic RGB uint8 inputs for the CLIP-preprocess goldens (recipe duplicated in tests/conftest.py)."""
    hh, ww = {"sq500": (500, 500), "sq336": (336, 336), "land": (375, 500), "small": (64, 48)}[name]
    rng = np.random.default_rng(1800 + hh + ww)
    im = rng.integers(0, 256, (hh, ww, 3), dtype=np.uint8)
    if name == "sq500":       # smooth content as well as noise
        yy, xx = np.mgrid[0:hh, 0:ww]
        im[..., 0] = (127 + 120 * np.sin(xx / 17.0) * np.cos(yy / 23.0)).astype(np.uint8)
    return im


def clip_digest(out: np.ndarray) -> dict:
    """Small stand-in for a 1.35 MB float32 output: a strided sub-grid (exact values) + order-free checksums
    of ALL values (the float32 bit patterns summed as integers, plain and position-weighted)."""
    bits = out.view(np.uint32).astype(np.uint64)
    pos = (np.arange(out.size, dtype=np.uint64) % np.uint64(65521)) + np.uint64(1)
    return {"sub": out[:, ::7, ::5].copy(), "sum_bits": np.array(bits.sum(dtype=np.uint64)),
            "wsum_bits": np.array((bits.ravel() * pos).sum(dtype=np.uint64))}


def make_clip_goldens():
    """"next" row 3: warped image -> CLIP tensor.  Third-party code only (run BEFORE the stubs are installed):
    Pillow BICUBIC + the HF CLIP image processor (PIL backend) configured like LLaVA-1.5's."""
    from transformers.models.clip.image_processing_pil_clip import CLIPImageProcessorPil
    import transformers
    clip = {}
    proc = CLIPImageProcessorPil(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336})
    for name in ("sq500", "sq336", "land", "small"):
        out = proc.preprocess([clip_input(name)], return_tensors="np")["pixel_values"][0]
        for k, v in clip_digest(out).items():
            clip[f"{name}_{k}"] = v
    rng = np.random.default_rng(18)
    bic = rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)
    clip["bic_img"] = bic
    for (w, h) in [(30, 20), (100, 90), (56, 17)]:
        clip[f"bic_{w}x{h}"] = np.array(Image.fromarray(bic).resize((w, h), Image.BICUBIC))
    clip["versions"] = np.array([Image.__version__, transformers.__version__])
    np.savez_compressed(os.path.join(OUT, "clip_preprocess.npz"), **clip)


def probe_cases():
    """Inputs of the "next" row 4 goldens: (name, query [B,H,q,D], key [B,Hkv,kv,D], left pads, slice starts)."""
    g = torch.Generator().manual_seed(19)
    B, H, Hkv, D, ntok = 2, 4, 2, 64, 24
    cases = []
    for name, q, kv in (("prefill", 48, 48), ("decode", 1, 49)):
        query = torch.randn(B, H, q, D, generator=g)
        key = torch.randn(B, Hkv, kv, D, generator=g)
        cases.append((name, query, key, [0, 3], [5, 8], ntok))
    return cases


def make_probe_golden(eager_attention_forward, llava):
    """"next" row 4: what the reference's hook sees when the target layer runs HF's eager attention
    (third-party: transformers' eager_attention_forward, imported before the stubs) and what
    BatchMaskHookLogger._process_attention (the reference) makes of it."""
    import transformers
    out = {"versions": np.array([transformers.__version__, torch.__version__])}
    for name, query, key, pads, starts, ntok in probe_cases():
        B, H, q, D = query.shape
        Hkv, kv = key.shape[1], key.shape[2]
        module = types.SimpleNamespace(num_key_value_groups=H // Hkv, training=False)
        scaling = D ** -0.5
        for tag, dt in (("f32", torch.float32), ("f16", torch.float16)):
            i = torch.arange(kv - q, kv)[:, None]
            j = torch.arange(kv)[None, :]
            allowed = (j <= i)[None, None] & (j[None, None] >= torch.tensor(pads)[:, None, None, None])
            mask = torch.zeros(B, 1, q, kv, dtype=dt).masked_fill(~allowed, torch.finfo(dt).min)
            qd, kd = query.to(dt), key.to(dt)
            vd = torch.zeros_like(kd)
            _, probs = eager_attention_forward(module, qd, kd, vd, mask, scaling=scaling)
            logger = llava.BatchMaskHookLogger(model=None, device="cpu", layer_index=20)
            logger.set_batch_image_token_ranges(starts, [s + ntok for s in starts])
            logger._process_attention(probs)
            out[f"{name}_{tag}_probs_last"] = probs[:, :, -1, :].numpy()
            out[f"{name}_{tag}_step"] = logger.step_attentions[0].numpy()
        out[f"{name}_q_last"] = query[:, :, -1, :].numpy()
        out[f"{name}_key"] = key.numpy()
        out[f"{name}_pads"] = np.array(pads)
        out[f"{name}_starts"] = np.array(starts)
        out[f"{name}_ntok"] = np.array(ntok)
        out[f"{name}_scaling"] = np.array(scaling)
    np.savez_compressed(os.path.join(OUT, "attn_probe.npz"), **out)


def config1_inputs():
    """BASELINE configs[0] / SURVEY 8d "config 1": one 336x336x3 uint8 RGB image and one 24x24 float32 attention
    map normalised to sum 1.  Recipe shared with the tests (the image itself is not stored)."""
    img = np.random.default_rng(0).integers(0, 256, (336, 336, 3), dtype=np.uint8)
    att = np.random.default_rng(1).random((24, 24))
    att = (att / att.sum()).astype(np.float32)
    return img, att


def make_config1_golden(llava, new_method):
    """The reference's own CPU-runnable case end to end (main.py / main_batched.py inner loop): revise_mask ->
    toImg -> PIL LANCZOS to the image size (blend_mask's mask branch, llava.py:241-253) -> warp_image_by_attention
    with the identity transform, at the input size and at the reference default 500x500; the maps handed to
    cv2.remap are captured by the stub."""
    img, att = config1_inputs()
    mask = llava.revise_mask(torch.from_numpy(att).float(), kernel_size=3, enhance_coe=10).detach().cpu()
    mask_pil = llava.toImg(mask.reshape(1, 24, 24))
    mota = llava.invtrans(mask_pil, Image.fromarray(img), method=Image.LANCZOS)
    mota_u8 = np.array(mota.convert("L"))
    out = {"att": att, "mota": mota_u8, "img_sum": np.array(int(img.sum()))}
    new_method.set_transform_function("identity", 1.0, 1.0, False)
    bgr = img[..., ::-1].copy()
    for (w, h) in [(336, 336), (500, 500)]:
        new_method.warp_image_by_attention(bgr, mota_u8, w, h)
        out[f"mx_{w}"] = CAPTURED["map_x"][0].copy()
        out[f"my_{h}"] = CAPTURED["map_y"][:, 0].copy()
    np.savez_compressed(os.path.join(OUT, "config1.npz"), **out)


MAIN_BATCHED_LOOP_WH = [(683, 1024), (500, 375), (333, 500), (1024, 768), (640, 427)]      # W x H as PIL reports them


def main_batched_loop_inputs():
    """The inputs of main_batched_loop.npz: five RGB images of different sizes (recipe, not stored: smooth content + noise so
    that a wrong map shows) and their 24 x 24 attention maps -- four seeded maps and, for the LAST image, the constant 1/576 map
    of the driver's OOM fallback (main_batched.py:231).  Recipe shared with the tests (tests/conftest.py)."""
    imgs, atts = [], []
    for i, (w, h) in enumerate(MAIN_BATCHED_LOOP_WH):
        rng = np.random.default_rng(2600 + i)
        yy, xx = np.mgrid[0:h, 0:w]
        im = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        im[..., 1] = (127 + 120 * np.sin(xx / (11.0 + i)) * np.cos(yy / (17.0 + 2 * i))).astype(np.uint8)
        imgs.append(im)
        a = rng.random((24, 24)) ** 3
        a[5 + i:8 + i, 9:12] += 4.0
        atts.append((a / a.sum()).astype(np.float32))
    atts[-1] = (torch.ones(24, 24) / 576).numpy()
    return imgs, atts


def make_main_batched_loop_golden(llava, new_method):
    """The composed per-sample loop of the reference's batched driver on DIFFERENTLY sized images, run through the reference's
    own functions in the driver's order (main_batched.py:243-287): blend_mask (llava.py:240-270: revise_mask -> toImg ->
    invtrans(LANCZOS) to image.size) -> np.array(mota.convert('L')) -> save_warped_image(image_path=<PIL image>, att_map=mota_np,
    width=500, height=500, transform="identity", ...) (new_method.py:405-506), whose cv2.remap call the stub records.
    Stored per image: the uint8 `mota` and the two float32 maps.  The last image gets the constant 1/576 map: revise_mask
    turns it into NaN (0 / 0), and the fixture records what torch's `.mul(255).byte()` makes of NaN on the reference's CPU path."""
    imgs, atts = main_batched_loop_inputs()
    out = {"sizes_wh": np.array(MAIN_BATCHED_LOOP_WH), "torch_version": np.array(torch.__version__),
           "pillow_version": np.array(Image.__version__)}
    for i, (im, att) in enumerate(zip(imgs, atts)):
        pil = Image.fromarray(im)
        with np.errstate(all="ignore"):
            _, mota_mask = llava.blend_mask(pil, torch.from_numpy(att), 10, 3, Image.LANCZOS, 0)
            mota_np = np.array(mota_mask.convert("L"))
            CAPTURED.clear()
            ok = new_method.save_warped_image(image_path=pil, att_map=mota_np, original_image_save_path=None,
                                              masked_overlay_save_path=None, output_path="/dev/null", vis_path=None,
                                              width=500, height=500, transform="identity", exp_scale=1.0, exp_divisor=1.0,
                                              apply_inverse=False)
        assert ok is True and mota_np.shape == im.shape[:2]
        assert np.all(CAPTURED["map_x"] == CAPTURED["map_x"][0:1]) and np.all(CAPTURED["map_y"] == CAPTURED["map_y"][:, 0:1])
        out[f"att_{i}"] = att
        out[f"mota_{i}"] = mota_np
        out[f"mx_{i}"] = CAPTURED["map_x"][0].copy()
        out[f"my_{i}"] = CAPTURED["map_y"][:, 0].copy()
        out[f"img_sum_{i}"] = np.array(int(im.sum()))
    np.savez_compressed(os.path.join(OUT, "main_batched_loop.npz"), **out)
    print("main_batched_loop.npz", os.path.getsize(os.path.join(OUT, "main_batched_loop.npz")) / 1024, "KiB; NaN map -> mota values",
          np.unique(out[f"mota_{len(imgs) - 1}"]))


def make_marginalnet_tail_golden(model):
    """"next" row 1 tail: the tensors around the reference MarginalNet's text pooling and FiLM + axis means
    (MN/model.py:73-88), captured with hooks on the reference module: masked token mean (input of txt_pool),
    film output, the map before FiLM (proj_v output, bilinearly up-sampled by the reference when sizes differ),
    vx / vy (inputs of head_x / head_y)."""
    out = {}
    for name, (hv, wv, H, W) in {"same": (24, 24, 24, 24), "up": (12, 10, 24, 32)}.items():
        torch.manual_seed(23)
        net = model.MarginalNet(d_vis_in=20, d_txt_in=40, hidden=16).eval()
        fmap = torch.randn(2, 20, hv, wv); ttok = torch.randn(2, 9, 40)
        tmask = torch.ones(2, 9, 1); tmask[1, 3:] = 0
        cap = {}
        hs = [net.txt_pool.register_forward_pre_hook(lambda m, a: cap.__setitem__("tmean", a[0].detach().clone())),
              net.film.register_forward_hook(lambda m, a, o: cap.__setitem__("gamma_beta", o.detach().clone())),
              net.proj_v.register_forward_hook(lambda m, a, o: cap.__setitem__("v_proj", o.detach().clone())),
              net.head_x.register_forward_pre_hook(lambda m, a: cap.__setitem__("vx", a[0].detach().clone())),
              net.head_y.register_forward_pre_hook(lambda m, a: cap.__setitem__("vy", a[0].detach().clone()))]
        with torch.no_grad():
            px, py = net(fmap, H, W, ttok, tmask)
            v_up = torch.nn.functional.interpolate(cap["v_proj"], size=(H, W), mode="bilinear", align_corners=False)
        for h in hs:
            h.remove()
        out.update({f"{name}_ttok": ttok.numpy(), f"{name}_tmask": tmask.numpy(), f"{name}_tmean": cap["tmean"].numpy(),
                    f"{name}_gamma_beta": cap["gamma_beta"].numpy(), f"{name}_v": v_up.numpy(),
                    f"{name}_vx": cap["vx"].numpy(), f"{name}_vy": cap["vy"].numpy(),
                    f"{name}_fmap": fmap.numpy(), f"{name}_px": px.numpy(), f"{name}_py": py.numpy(),
                    f"{name}_HW": np.array([H, W])})
        for k, v in net.state_dict().items():
            out[f"{name}_sd|{k}"] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "marginalnet_tail.npz"), **out)


def make_marginalnet_full_golden(model):
    """BASELINE configs[4] at its stated size: the reference MarginalNet(1024, 4096, hidden=256)
    (MN/model.py:25-53, config.py:26) forward at B=4 on seeded weights / inputs (recipes in tests/conftest.py, so the
    11 MB of weights are not committed).  Stored: (px, py), the logits' inputs vx / vy as a strided sub-grid, and
    float64 checksums of every hooked tensor and of the state_dict."""
    sys.path.insert(0, os.path.dirname(OUT))
    from conftest import marginalnet_full_state, marginalnet_full_inputs
    net = model.MarginalNet(d_vis_in=1024, d_txt_in=4096, hidden=256).eval()
    sd = marginalnet_full_state({k: tuple(v.shape) for k, v in net.state_dict().items()})
    net.load_state_dict(sd)
    fmap, ttok, tmask = marginalnet_full_inputs(4)
    cap = {}
    hs = [net.txt_pool.register_forward_pre_hook(lambda m, a: cap.__setitem__("tmean", a[0].detach().clone())),
          net.film.register_forward_hook(lambda m, a, o: cap.__setitem__("gamma_beta", o.detach().clone())),
          net.head_x.register_forward_pre_hook(lambda m, a: cap.__setitem__("vx", a[0].detach().clone())),
          net.head_y.register_forward_pre_hook(lambda m, a: cap.__setitem__("vy", a[0].detach().clone()))]
    with torch.no_grad():
        px, py = net(fmap, 24, 24, ttok, tmask)
    for h in hs:
        h.remove()
    out = {"px": px.numpy(), "py": py.numpy(), "vx_sub": cap["vx"][:, ::16, ::3].numpy(),
           "vy_sub": cap["vy"][:, ::16, ::3].numpy(), "tmean_sub": cap["tmean"][:, ::64].numpy(),
           "gamma_beta_sub": cap["gamma_beta"][:, ::32].numpy(),
           "sum_vx": np.array(cap["vx"].double().sum().item()), "sum_vy": np.array(cap["vy"].double().sum().item()),
           "sum_tmean": np.array(cap["tmean"].double().sum().item()),
           "sd_checksum": np.array(sum(v.double().sum().item() for v in sd.values())),
           "n_params": np.array(sum(v.numel() for v in sd.values()))}
    np.savez_compressed(os.path.join(OUT, "marginalnet_full.npz"), **out)


def make_random_cases(new_method, ckpt, llava):
    """Randomised, partly hostile cases through the reference (complements the hand-picked goldens; the long-form
    version of this is fuzz_reference.py).  Inputs are smooth + sparse so that the file compresses; stored: inputs and
    the reference's outputs only."""
    rng = np.random.default_rng(4242)
    out = {}
    names = []
    # ---- A13: maps of warp_image_by_attention (bit-exact target) ----
    for i in range(48):
        h, w = int(rng.integers(2, 70)), int(rng.integers(2, 90)); nw, nh = int(rng.integers(1, 300)), int(rng.integers(1, 300))
        kind = ["u8", "f32", "f64"][i % 3]
        yy, xx = np.mgrid[0:h, 0:w]
        base = 0.5 + 0.5 * np.sin(xx / (3.0 + i % 7)) * np.cos(yy / (2.0 + i % 5))
        base[rng.integers(0, h), rng.integers(0, w)] += 4.0
        if kind == "u8": att = np.clip(base * 60, 0, 255).astype(np.uint8)
        elif kind == "f32": att = (base - 0.3).astype(np.float32)
        else: att = (base - 0.6).astype(np.float64)
        r = i % 12
        if r == 3: att[:] = 0
        elif r == 5: att[: h // 2] = 0
        elif r == 7 and kind != "u8": att[h // 2, w // 2] = np.nan
        elif r == 9 and kind != "u8": att[0, 0] = np.inf
        elif r == 11 and kind != "u8": att[:] = -1.0
        tr = ["identity", "square", "sqrt", "exp", "log", "bogus"][i % 6]; inv = bool((i // 6) % 2)
        es, ed = [(1.0, 1.0), (0.01, 2.0), (2.0, 50.0)][i % 3]
        new_method.set_transform_function(tr, es, ed, inv)
        with np.errstate(all="ignore"):
            new_method.warp_image_by_attention(np.zeros((h, w, 3), np.uint8), att, nw, nh)
        key = f"a13_{i}"
        out[key + "_att"] = att; out[key + "_mx"] = CAPTURED["map_x"][0].copy(); out[key + "_my"] = CAPTURED["map_y"][:, 0].copy()
        out[key + "_par"] = np.array([nw, nh, ["identity", "square", "sqrt", "exp", "log", "bogus"].index(tr), int(inv), es, ed])
        names.append(key)
    # ---- A9 + A11: cdf_from_density -> maps of warp_from_cdf_torch (bit-exact target) ----
    for i in range(40):
        H, W = int(rng.integers(24, 160)), int(rng.integers(24, 200))
        osz = None if i % 3 == 0 else (int(rng.integers(2, 260)), int(rng.integers(2, 260)))
        p = (0.2 + np.abs(np.sin(np.arange(W) / (2.0 + i % 9)))).astype(np.float32); q = (0.1 + np.abs(np.cos(np.arange(H) / (3.0 + i % 4)))).astype(np.float32)
        r = i % 10
        if r == 1: p[W // 4: W // 2] = 0
        elif r == 3: p[:] = 0
        elif r == 5: q[: H // 3] = 0; q[-3:] = 0
        elif r == 7: p[W // 2] = np.nan
        elif r == 9: p = (p ** 12).astype(np.float32)
        Fx = ckpt.cdf_from_density(torch.from_numpy(p[None].copy())); Fy = ckpt.cdf_from_density(torch.from_numpy(q[None].copy()))
        ckpt.warp_from_cdf_torch(torch.zeros(1, 1, H, W), Fx, Fy, osz)
        key = f"a11_{i}"
        out[key + "_p"] = p; out[key + "_q"] = q; out[key + "_Fx"] = Fx.numpy()[0]; out[key + "_Fy"] = Fy.numpy()[0]
        out[key + "_mx"] = CAPTURED["map_x"][0].copy(); out[key + "_my"] = CAPTURED["map_y"][:, 0].copy()
        out[key + "_out"] = np.array(osz if osz else (H, W))
        names.append(key)
    # ---- A3: revise_mask incl. a constant map ----
    for i in range(12):
        yy, xx = np.mgrid[0:24, 0:24]
        m = (0.5 + 0.45 * np.sin(xx / (1.5 + i)) * np.cos(yy / (2.5 + i % 3))).astype(np.float32)
        if i == 4: m[:] = 0.25
        ks, coe = [(3, 10.0), (5, 4.0), (1, 30.0)][i % 3]
        with np.errstate(all="ignore"):
            rev = llava.revise_mask(torch.from_numpy(m.copy()), kernel_size=ks, enhance_coe=coe).detach().numpy().reshape(24, 24)
        key = f"a3_{i}"
        out[key + "_m"] = m; out[key + "_rev"] = rev; out[key + "_par"] = np.array([ks, coe])
        names.append(key)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "random_cases.npz"), **out)
    print("random_cases.npz", os.path.getsize(os.path.join(OUT, "random_cases.npz")) / 1024, "KiB,", len(names), "cases")


def make_opencv_golden():
    """`--with-opencv`: when the REAL OpenCV is importable (it is not in the build container: no network), record
    cv2.remap(INTER_LINEAR, BORDER_REPLICATE) / cv2.resize(INTER_LINEAR) outputs on seeded inputs as remap_cv2.npz -- the
    fixture that pins oracle.remap_bilinear(mode="cv2") and every HIP resample kernel to OpenCV itself on machines without
    it (tests/test_oracle_golden.py::test_remap_cv2_fixture_if_present, the -m gpu twin).  Needs no reference import."""
    try:
        import cv2
    except ImportError:
        print("--with-opencv: cv2 is not importable here; nothing written (tests/test_oracle_vs_opencv.py skips the same way)")
        return
    if not hasattr(cv2, "getBuildInformation"):
        print("--with-opencv: `cv2` is a stub, not OpenCV; nothing written")
        return
    rng = np.random.default_rng(2024)
    out, names = {}, []
    for (H, W, Ho, Wo) in ((48, 64, 52, 72), (336, 336, 500, 500), (100, 683, 90, 500), (33, 47, 40, 31)):
        for dt in (np.uint8, np.float32, np.float64):
            for C in (1, 3, 4):
                img = rng.integers(0, 256, (H, W, C), dtype=np.uint8) if dt == np.uint8 else rng.random((H, W, C)).astype(dt)
                mx = np.sort(rng.random(Wo).astype(np.float32) * (W + 2) - 1)
                my = np.sort(rng.random(Ho).astype(np.float32) * (H + 2) - 1)
                mx[:4] = np.array([0.015625, 0.046875, -7.25, W - 0.015625], np.float32)
                X, Y = np.meshgrid(mx, my)
                res = cv2.remap(img if C > 1 else img[:, :, 0], X, Y, interpolation=cv2.INTER_LINEAR, borderMode=cv2.BORDER_REPLICATE)
                key = f"{H}x{W}x{C}_{Ho}x{Wo}_{np.dtype(dt).name}"
                out[f"{key}|img"] = img if C > 1 else img[:, :, 0]
                out[f"{key}|mx"] = mx; out[f"{key}|my"] = my; out[f"{key}|out"] = res
                names.append(key)
    out["cases"] = np.array(names)
    out["opencv_build"] = np.array(f"OpenCV {cv2.__version__}")
    np.savez_compressed(os.path.join(OUT, "remap_cv2.npz"), **out)
    print("remap_cv2.npz", os.path.getsize(os.path.join(OUT, "remap_cv2.npz")) / 1024, "KiB,", len(names), "cases,", cv2.__version__)


def main():
    if "--with-opencv" in sys.argv:
        make_opencv_golden()
        return
    if "--only-random" in sys.argv:
        _install_stubs()
        torch.set_num_threads(1)
        new_method = _load("ref_new_method", os.path.join(AGW, "new_method.py"))
        ckpt = _load("ref_checkpoint_utils", os.path.join(MN, "checkpoint_utils.py"))
        sys.path.insert(0, os.path.join(AGW, "attention_extraction"))
        llava = _load("ref_llava", os.path.join(AGW, "attention_extraction", "llava.py"))
        make_random_cases(new_method, ckpt, llava)
        return
    if "--only-loop" in sys.argv:
        _install_stubs()
        torch.set_num_threads(1)
        new_method = _load("ref_new_method", os.path.join(AGW, "new_method.py"))
        sys.path.insert(0, os.path.join(AGW, "attention_extraction"))
        llava = _load("ref_llava", os.path.join(AGW, "attention_extraction", "llava.py"))
        make_main_batched_loop_golden(llava, new_method)
        return
    if "--only-mnfull" in sys.argv:
        _install_stubs()
        torch.set_num_threads(1)
        make_marginalnet_full_golden(_load("ref_model", os.path.join(MN, "model.py")))
        return
    if not any(a.startswith("--only-") for a in sys.argv):
        make_clip_goldens()
    from transformers.models.llama.modeling_llama import eager_attention_forward
    _install_stubs()
    torch.manual_seed(0)
    torch.set_num_threads(1)

    new_method = _load("ref_new_method", os.path.join(AGW, "new_method.py"))
    ckpt = _load("ref_checkpoint_utils", os.path.join(MN, "checkpoint_utils.py"))
    model = _load("ref_model", os.path.join(MN, "model.py"))
    sys.path.insert(0, os.path.join(AGW, "attention_extraction"))
    llava = _load("ref_llava", os.path.join(AGW, "attention_extraction", "llava.py"))

    make_config1_golden(llava, new_method)
    if "--only-config1" in sys.argv:
        return
    make_main_batched_loop_golden(llava, new_method)
    make_marginalnet_tail_golden(model)
    if "--only-mntail" in sys.argv:
        return
    make_marginalnet_full_golden(model)
    if "--only-mntail" not in sys.argv:
        make_probe_golden(eager_attention_forward, llava)
    if "--only-probe" in sys.argv:
        return

    # ---------------- A1 / A2: attention aggregation -----------------------
    g = torch.Generator().manual_seed(11)
    B, T, heads, kv = 3, 4, 32, 640
    starts = [35, 37, 41]
    ends = [s + 576 for s in starts]
    steps_in = []
    logger = llava.BatchMaskHookLogger(model=None, device="cpu", layer_index=20)
    logger.set_batch_image_token_ranges(starts, ends)
    for t in range(T):
        q = 5 if t == 0 else 1
        a = torch.softmax(torch.randn(B, heads, q, kv, generator=g), dim=-1)
        steps_in.append(a.numpy())
        logger._process_attention(a)
    step_out = [s.numpy() for s in logger.step_attentions]
    final = torch.stack(logger.finalize_batch()).numpy()
    # fp16 variant (what LLaVA actually produces)
    logger16 = llava.BatchMaskHookLogger(model=None, device="cpu", layer_index=20)
    logger16.set_batch_image_token_ranges(starts, ends)
    for a in steps_in:
        logger16._process_attention(torch.from_numpy(a).half())
    step_out16 = [s.numpy() for s in logger16.step_attentions]
    final16 = torch.stack(logger16.finalize_batch()).numpy()
    empty = llava.BatchMaskHookLogger(model=None, device="cpu")
    empty.set_batch_image_token_ranges([1, 2], [577, 578])
    empty_out = torch.stack(empty.finalize_batch()).numpy()
    np.savez_compressed(os.path.join(OUT, "attn_reduce.npz"),
                        starts=np.array(starts), ends=np.array(ends),
                        **{f"step_in_{t}": steps_in[t][:, :, -1:, :] for t in range(T)},
                        **{f"step_out_{t}": step_out[t] for t in range(T)},
                        **{f"step_out16_{t}": step_out16[t] for t in range(T)},
                        final=final, final16=final16, empty=empty_out)

    # ---------------- A3: mask post-processing ------------------------------
    g = torch.Generator().manual_seed(12)
    masks = torch.rand(4, 24, 24, generator=g) / 576.0
    masks[1, 5:8, 9:12] *= 100.0                     # peaked attention
    masks[3] = final_t = torch.from_numpy(final[0])  # a realistic aggregated map
    revised = torch.stack([llava.revise_mask(m.float(), kernel_size=3, enhance_coe=10) for m in masks])
    revised = revised.detach().reshape(4, 24, 24).numpy()
    revised5 = llava.revise_mask(masks[0].float(), kernel_size=5, enhance_coe=4).detach().reshape(24, 24).numpy()
    # A4: ToPILImage truncation + Pillow LANCZOS (real Pillow, third-party)
    u8 = np.stack([np.array(llava.toImg(torch.from_numpy(r).reshape(1, 24, 24))) for r in revised])
    lanczos = {}
    for (w, h) in [(336, 336), (500, 375), (1024, 1024), (24, 48), (17, 24)]:
        lanczos[f"lanczos_{w}x{h}"] = np.stack(
            [np.array(Image.fromarray(m, mode="L").resize((w, h), Image.LANCZOS)) for m in u8[:2]])
    np.savez_compressed(os.path.join(OUT, "mask_postproc.npz"), masks=masks.numpy(), revised=revised,
                        mask0_k5_c4=revised5, u8=u8, pillow_version=np.array(Image.__version__), **lanczos)

    # ---------------- A5/A6/A7 ---------------------------------------------
    # Big inputs are NOT stored: the tests regenerate them with the same numpy
    # Generator recipe (PCG64 streams are stable across numpy versions).
    pool = {}
    for S in (336, 512, 1024):
        A = pool_input(S)
        P = torch.nn.functional.adaptive_avg_pool2d(torch.from_numpy(A), (24, 24))
        pool[f"P_{S}"] = P.numpy()
        px, py = ckpt.gt_marginals(P)
        pool[f"px_{S}"] = px.numpy(); pool[f"py_{S}"] = py.numpy()
    g = torch.Generator().manual_seed(13)
    # non-square + negative values for gt_marginals at "full resolution"
    Afull = torch.randn(2, 1, 40, 56, generator=g)
    pxf, pyf = ckpt.gt_marginals(Afull)
    logits = torch.randn(5, 24, generator=g) * 4
    logits[1, 3] = float("nan"); logits[2, 7] = float("inf"); logits[3, :] = -float("inf")
    sm = model.safe_softmax(logits.clone(), dim=1, eps=1e-6)
    np.savez_compressed(os.path.join(OUT, "pool_marginals.npz"), **pool,
                        Afull=Afull.numpy(), pxf=pxf.numpy(), pyf=pyf.numpy(),
                        logits=logits.numpy(), safe_softmax=sm.numpy())

    # ---------------- A8/A9/A10 --------------------------------------------
    g = torch.Generator().manual_seed(14)
    y = torch.softmax(torch.randn(3, 24, generator=g) * 2, dim=1)
    a8 = {"y": y.numpy()}
    for L in (336, 500, 512, 1024):
        x = ckpt.upsample_pdf_right_inverse(y, L)
        a8[f"x_{L}"] = x.numpy()
        a8[f"cdf_{L}"] = ckpt.cdf_from_density(x.clamp_min(0)).numpy()
    a8["x1d_336"] = ckpt.upsample_pdf_right_inverse(y[0], 336).numpy()
    a8["x3d_336"] = ckpt.upsample_pdf_right_inverse(y.reshape(1, 3, 24), 336).numpy()
    p_bad = torch.rand(3, 50, generator=g)
    p_bad[0, 3] = float("nan"); p_bad[1, 4] = -2.0; p_bad[1, 9] = float("inf"); p_bad[2] = 0.0
    a8["p_bad"] = p_bad.numpy()
    a8["cdf_bad"] = ckpt.cdf_from_density(p_bad.clone()).numpy()
    Fc = torch.cumsum(torch.softmax(torch.randn(3, 24, generator=g), dim=1), dim=1)
    Fc[1, 5:9] = Fc[1, 5]                      # plateau
    Fc[2, 10] = float("nan"); Fc[2, 3] = Fc[2, 2] - 0.05   # nan + decreasing
    a8["F24"] = Fc.numpy()
    a8["msi"] = ckpt._make_strictly_increasing(Fc.clone()).numpy()
    for L in (336, 1024):
        a8[f"resample_{L}"] = ckpt.resample_cdf(Fc.clone(), L).numpy()
    np.savez_compressed(os.path.join(OUT, "pdf_cdf.npz"), **a8)

    # ---------------- A11: maps from CDFs (torch variant) -------------------
    g = torch.Generator().manual_seed(15)
    a11 = {}
    cases = {
        "sq336": (336, 336, None),
        "rect": (48, 64, (40, 72)),
        "to500": (336, 336, (500, 500)),
        "sq1024": (1024, 1024, None),
    }
    for name, (H, W, out_size) in cases.items():
        px = torch.softmax(torch.randn(2, 24, generator=g) * 2, dim=1)
        py = torch.softmax(torch.randn(2, 24, generator=g) * 2, dim=1)
        Fx = ckpt.cdf_from_density(ckpt.upsample_pdf_right_inverse(px, W).clamp_min(0))
        Fy = ckpt.cdf_from_density(ckpt.upsample_pdf_right_inverse(py, H).clamp_min(0))
        img = torch.zeros(2, 3, H, W)
        mxs, mys = [], []
        for b in range(2):
            ckpt.warp_from_cdf_torch(img[b:b + 1], Fx[b:b + 1], Fy[b:b + 1], out_size)
            mxs.append(CAPTURED["map_x"][0].copy()); mys.append(CAPTURED["map_y"][:, 0].copy())
            assert np.all(CAPTURED["map_x"] == CAPTURED["map_x"][0:1])       # separable
            assert np.all(CAPTURED["map_y"] == CAPTURED["map_y"][:, 0:1])
        a11[f"{name}_Fx"] = Fx.numpy(); a11[f"{name}_Fy"] = Fy.numpy()
        a11[f"{name}_mx"] = np.stack(mxs); a11[f"{name}_my"] = np.stack(mys)
        a11[f"{name}_out"] = np.array(out_size if out_size else (H, W))
    # ties: zero-density stretches -> flat CDF -> tie-break ramp branch
    p = torch.rand(2, 64, generator=g); p[0, 10:30] = 0; p[1, :8] = 0; p[1, 60:] = 0
    Ft = ckpt.cdf_from_density(p)
    img = torch.zeros(2, 1, 64, 64)
    mxs, mys = [], []
    for b in range(2):
        ckpt.warp_from_cdf_torch(img[b:b + 1], Ft[b:b + 1], Ft[b:b + 1].flip(0), (80, 96))
        mxs.append(CAPTURED["map_x"][0].copy()); mys.append(CAPTURED["map_y"][:, 0].copy())
    a11["ties_F"] = Ft.numpy(); a11["ties_mx"] = np.stack(mxs); a11["ties_my"] = np.stack(mys)
    a11["ties_out"] = np.array((80, 96))
    np.savez_compressed(os.path.join(OUT, "maps_from_cdf.npz"), **a11)

    # ---------------- A13: maps from attention (numpy float64 variant) -----
    rng = np.random.default_rng(16)
    a13 = {}
    att_u8 = np.array(Image.fromarray(u8[1], mode="L").resize((336, 336), Image.LANCZOS))
    att_f = rng.random((60, 84)).astype(np.float32)
    att_f[20:30, 40:60] *= 20
    att_neg = (rng.standard_normal((32, 48))).astype(np.float64)
    a13["att_u8"] = att_u8; a13["att_f"] = att_f; a13["att_neg"] = att_neg
    a13["att_zero"] = np.zeros((24, 40), dtype=np.uint8)
    combos = []
    for aname, att in [("att_u8", att_u8), ("att_f", att_f), ("att_neg", att_neg), ("att_zero", a13["att_zero"])]:
        for tr in ["identity", "square", "sqrt", "exp", "log", "bogus"]:
            for inv in (False, True):
                for (nw, nh) in [(500, 500), (att.shape[1], att.shape[0])]:
                    if aname == "att_u8" and tr == "exp":
                        es, ed = 0.01, 2.0
                    else:
                        es, ed = 1.0, 1.0
                    new_method.set_transform_function(tr, es, ed, inv)
                    img = np.zeros(att.shape + (3,), dtype=np.uint8)
                    with np.errstate(all="ignore"):
                        new_method.warp_image_by_attention(img, att, nw, nh)
                    key = f"{aname}|{tr}|{int(inv)}|{nw}x{nh}|{es}|{ed}"
                    combos.append(key)
                    a13[f"mx|{key}"] = CAPTURED["map_x"][0].copy()
                    a13[f"my|{key}"] = CAPTURED["map_y"][:, 0].copy()
    a13["combos"] = np.array(combos)
    np.savez_compressed(os.path.join(OUT, "maps_from_attention.npz"), **a13)

    # ---------------- MarginalNet (next-row producer) -----------------------
    torch.manual_seed(17)
    net = model.MarginalNet(d_vis_in=32, d_txt_in=48, hidden=16).eval()
    fmap = torch.randn(2, 32, 24, 24); ttok = torch.randn(2, 7, 48)
    tmask = torch.ones(2, 7, 1); tmask[1, 4:] = 0
    with torch.no_grad():
        px, py = net(fmap, 24, 24, ttok, tmask)
    sd = {k: v.numpy() for k, v in net.state_dict().items()}
    np.savez_compressed(os.path.join(OUT, "marginalnet.npz"), fmap=fmap.numpy(), ttok=ttok.numpy(),
                        tmask=tmask.numpy(), px=px.numpy(), py=py.numpy(),
                        **{"sd|" + k: v for k, v in sd.items()})
    n_full = sum(p.numel() for p in model.MarginalNet(1024, 4096, 256).parameters())
    print("MarginalNet(1024,4096,256) params:", n_full)

    make_random_cases(new_method, ckpt, llava)

    tot = 0
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            sz = os.path.getsize(os.path.join(OUT, f)); tot += sz
            print(f"{f:32s} {sz/1024:8.1f} KiB")
    print("total", tot / 1024, "KiB")


if __name__ == "__main__":
    main()
