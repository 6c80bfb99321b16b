"""Dev tool: a few launches of ONE part of the path, for `rocprofv3 --kernel-trace --stats` / `--pmc` (put this program
itself after `--`: rocprofv3 ... -- python3 tools/prof.py <target> [args]).  AB_LIB=<path> profiles another build.

targets
  remap [B S] [peaked] [key=value ...]   float32 resample at BASELINE configs[2]; key=value: attwarp_debug_set overrides
  chain [B S So]             every kernel of the main_batched chain (pipeline.warp_from_masks), default 256 1024 500
  chain_step [B S So]        the one-launch chain step (pipeline.MaskChainStream pattern "fused")
  u8 [B S So [W]]            the integer uint8 cv2 resample alone, [B,S,W,3] -> [B,So,So,3] (default 256 1024 1024)
  ragged [B]                 the ragged chain (TextVQA-like size mix -> 500 x 500): five launches per batch, then the one-launch stream step
  attn                       attention reduce, float32 + float16 rows, bench shape
  steps                      attention reduce + axis_maps_from_steps at the bench shapes
  steps summarize <csv>      per-grid-size medians of a --kernel-trace CSV of the above
  fused [B S]                the one-launch float32 step (pipeline.OverlappedWarp over a ring), default 256 336
  axis                       A13 maps from uint8 attention
  clip                       clip_preprocess, 64 x 500 x 500 x 3 -> 336
  mn [B]                     MarginalNet(1024, 4096, 256) inference forwards
"""
import csv, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = sys.argv[1:]
target = args[0] if args else ""
if target == "steps" and len(args) > 2 and args[1] == "summarize":
    by = {}
    for r in csv.DictReader(open(args[2])):
        n = r["Kernel_Name"]
        if "axis_maps_from_steps" not in n and "attn_reduce_step" not in n:
            continue
        key = (n.split("(")[0][-60:], r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Grid_Size_Y"))
        by.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in sorted(by.items()):
        print(f"{k}: n={len(v)} median {statistics.median(v):.2f} us  min {min(v):.2f}  max {max(v):.2f}")
    sys.exit(0)
import torch
from attwarp_amd import _lib
if os.environ.get("AB_LIB"):
    _lib.LIB_PATH = os.environ["AB_LIB"]
from attwarp_amd import checkpoint_utils as cu, new_method as nm, pipeline, model
dev = torch.device("cuda:0")
ints = [int(a) for a in args[1:] if a.lstrip("-").isdigit()]

if target == "remap":
    B, S = (ints + [256, 1024])[:2] if len(ints) >= 2 else (256, 1024)
    kind = "peaked" if "peaked" in args else "uniform"
    g = torch.Generator(device=dev).manual_seed(1)
    sc = 0.02 if kind == "uniform" else 2.0
    px = torch.softmax(torch.randn(B, 24, device=dev, generator=g) * sc, 1)
    py = torch.softmax(torch.randn(B, 24, device=dev, generator=g) * sc, 1)
    mx, my = pipeline.axis_maps_from_pdf(px, py, (S, S))
    img = torch.rand((B, S, S, 3), device=dev); out = torch.empty_like(img)
    over = {k: int(v) for k, v in (a.split("=") for a in args[1:] if "=" in a)}      # attwarp_debug_set overrides, e.g. remap_noswz=8
    with _lib.debug_override(**over) if over else __import__("contextlib").nullcontext():
        for _ in range(5):
            cu.remap_separable(img, mx, my, channels_last=True, out=out)
elif target in ("chain", "chain_step"):
    B, S, So = ints[:3] if len(ints) >= 3 else (256, 1024, 500)
    if target == "chain":
        img8 = (torch.rand(B, S, S, 3, device=dev) * 255).to(torch.uint8)
        m24 = torch.rand(B, 24, 24, device=dev)
        for _ in range(6):
            pipeline.warp_from_masks(img8, m24, (So, So))
    else:
        n = 6
        imgs = [(torch.rand(B, S, S, 3, device=dev) * 255).to(torch.uint8) for _ in range(n)]
        msk = [torch.rand(B, 24, 24, device=dev) for _ in range(n)]
        mc = pipeline.MaskChainStream(imgs, msk, (So, So), pattern="fused")
        mc.prime()
        for _ in range(12):
            mc.step()
elif target == "ragged":
    B = ints[0] if ints else 32
    WH = [(1024, 768), (683, 1024), (1024, 1024), (500, 375), (333, 500), (640, 427)]
    g = torch.Generator(device=dev).manual_seed(B)
    ring = []
    for _ in range(6):
        rb = pipeline.RaggedBatch([torch.randint(0, 256, (h, w, 3), device=dev, dtype=torch.uint8, generator=g) for (w, h) in (WH[b % 6] for b in range(B))], (500, 500))
        rb.masks = torch.rand(B, 24, 24, device=dev, generator=g)
        ring.append(rb)
    for _ in range(3):
        pipeline.warp_from_masks_ragged(ring[0].images, ring[0].masks, (500, 500))
    st = pipeline.RaggedMaskChainStream(out_size=(500, 500))
    st.ring(ring); st.prime(); st.run(24); st.drain_ring()
elif target == "u8":
    # the integer uint8 cv2 resample alone (remap_rows_u8i_kernel), maps from a mildly peaked 24-bin PDF
    B, S, So = ints[:3] if len(ints) >= 3 else (256, 1024, 1024)
    W = ints[3] if len(ints) >= 4 else S                     # e.g. `u8 256 1024 500 683`: 683-pixel-wide rows (the unaligned form)
    g = torch.Generator(device=dev).manual_seed(1)
    px = torch.softmax(torch.randn(B, 24, device=dev, generator=g) * 0.3, 1)
    py = torch.softmax(torch.randn(B, 24, device=dev, generator=g) * 0.3, 1)
    mx, my = pipeline.axis_maps_from_pdf(px, py, (S, W), (So, So))
    img = torch.randint(0, 256, (B, S, W, 3), device=dev, dtype=torch.uint8, generator=g)
    out = torch.empty(B, So, So, 3, device=dev, dtype=torch.uint8)
    for _ in range(8):
        cu.remap_separable(img, mx, my, channels_last=True, out=out)
elif target == "attn":
    B = 256
    rows = torch.softmax(torch.randn(20, B, 32, 640, device=dev), -1)
    starts = (35 + torch.arange(B, device=dev) % 8).int()
    st = starts.repeat(20)
    r16 = rows.half()
    for _ in range(5):
        pipeline.attention_step_maps(rows, starts, 576, st)
        pipeline.attention_step_maps(r16, starts, 576, st)
elif target == "steps":
    for (B, S) in [(64, 336), (256, 336), (256, 1024), (8, 336)]:
        rows = torch.softmax(torch.randn(20, B, 32, 640, device=dev), -1)
        starts = (35 + torch.arange(B, device=dev) % 8).to(torch.int32)
        for _ in range(12):
            steps = pipeline.attention_step_maps(rows, starts)
            pipeline.axis_maps_from_attention_steps(steps, (S, S))
        torch.cuda.synchronize()
elif target == "fused":
    import bench
    B, S = ints[:2] if len(ints) >= 2 else (256, 336)
    st = bench.Step(B, S, dev, seed=5, mode="cv2", layout="hwc")
    ow = pipeline.OverlappedWarp([x[0] for x in st.sets], [x[1] for x in st.sets], st.starts, channels_last=True, mode="cv2")
    ow.prime(); ow.prime2()
    for _ in range(12):
        ow.step()
elif target == "axis":
    for (B, S, So) in [(256, 1024, 500), (64, 336, 500), (1, 336, 500)]:
        au8 = (torch.rand(B, S, S, device=dev) * 255).to(torch.uint8)
        for _ in range(5):
            nm.attention_axis_maps(au8, So, So, "identity")
        torch.cuda.synchronize()
elif target == "clip":
    w500 = (torch.rand(64, 500, 500, 3, device=dev) * 255).to(torch.uint8)
    for _ in range(10):
        pipeline.clip_preprocess(w500)
elif target == "mn":
    B = ints[0] if ints else 256
    net = model.MarginalNet(1024, 4096, 256).to(dev).eval()
    fmap = torch.randn(B, 1024, 24, 24, device=dev)
    tok = torch.randn(B, 32, 4096, device=dev); msk = (torch.rand(B, 32, 1, device=dev) > 0.3).float()
    with torch.no_grad():
        for _ in range(6):
            net(fmap, 24, 24, tok, msk)
else:
    raise SystemExit(__doc__)
torch.cuda.synchronize()
