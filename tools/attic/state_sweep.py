"""Dev tool: which (rows per block, XCD grouping, row blocks per workgroup) of the float32 resample wins in which box
state.  Cycles over the variants for `dur` seconds on one box (B=256 1024x1024x3 HWC, cv2), a torch.add over the same
bytes in every cycle as the calibration, rocm-smi clocks / power sampled by a background thread.
    python tools/state_sweep.py [dur_seconds] [kind] [S] [B] ["R3c4:remap_rows=3,remap_cpw=4;..."]"""
import json, os, statistics, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from attwarp_amd import checkpoint_utils as cu, _lib
from remap_bench import maps

dur = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
kind = sys.argv[2] if len(sys.argv) > 2 else "uniform"
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
B = int(sys.argv[4]) if len(sys.argv) > 4 else 256
layout = sys.argv[6] if len(sys.argv) > 6 else "hwc"
mode = sys.argv[7] if len(sys.argv) > 7 else "cv2"
dev = torch.device("cuda:0")
img = torch.rand((B, S, S, 3) if layout == "hwc" else (B, 3, S, S), device=dev); out = torch.empty_like(img); ref = torch.empty_like(img)
mx, my = maps(B, S, kind)

VARIANTS = {
    "R4 g0 (default)": {},
    "R3 g4": dict(remap_rows=3, remap_noswz=4),
    "R3 g0": dict(remap_rows=3),
    "R4 g4": dict(remap_rows=4, remap_noswz=4),
    "R4 g2": dict(remap_rows=4, remap_noswz=2),
    "R3 g8": dict(remap_rows=3, remap_noswz=8),
    "R2 g8": dict(remap_rows=2, remap_noswz=8),
    "R4 g0 cpw2": dict(remap_cpw=2),
    "R4 g0 cpw4": dict(remap_cpw=4),
    "R4 g0 cpw8": dict(remap_cpw=8),
    "R3 g4 cpw4": dict(remap_rows=3, remap_noswz=4, remap_cpw=4),
    "R3 g0 cpw4": dict(remap_rows=3, remap_cpw=4),
    "R2 g0 cpw8": dict(remap_rows=2, remap_cpw=8),
    "R4 plain cpw4": dict(remap_noswz=1, remap_cpw=4),
    "R4 g0 nt": dict(remap_nt=1),
}

if len(sys.argv) > 5:
    VARIANTS = {"default": {}}
    for item in sys.argv[5].split(";"):
        name, kv = item.split(":")
        VARIANTS[name] = {k: int(v) for k, v in (a.split("=") for a in kv.split(",") if a)}

def run(over, n):
    ts = []
    with _lib.debug_override(**over):
        for _ in range(n):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); cu.remap_separable(img, mx, my, mode=mode, channels_last=(layout == "hwc"), out=out); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return ts

cu.remap_separable(img, mx, my, mode=mode, channels_last=(layout == "hwc"), out=ref)
for name, over in VARIANTS.items():
    out.zero_(); run(over, 1)
    assert torch.equal(out, ref), name
print("all variants bit-identical", flush=True)

smi_log, stop = [], False
def smi():
    while not stop:
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=10)
            d = json.loads(r.stdout); c = d[sorted(d)[0]]
            keep = {k: v for k, v in c.items() if any(s in k.lower() for s in ("sclk", "mclk", "fclk", "socclk", "power", "junction", "memory)"))}
            smi_log.append((time.time(), keep))
        except Exception as e:
            smi_log.append((time.time(), {"err": str(e)[:80]}))
        time.sleep(2.0)
th = threading.Thread(target=smi, daemon=True); th.start()

t_start = time.time()
hist = {k: [] for k in list(VARIANTS) + ["torch.add"]}
cyc = 0
while time.time() - t_start < dur:
    for name, over in VARIANTS.items():
        ts = run(over, 6)
        hist[name].append((time.time() - t_start, statistics.median(ts)))
    ts = []
    for _ in range(6):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); torch.add(img, 1.0, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    hist["torch.add"].append((time.time() - t_start, statistics.median(ts)))
    cyc += 1
    if cyc % 5 == 1:
        print(f"t={time.time()-t_start:6.1f}s " + " | ".join(f"{k.split(' (')[0]} {v[-1][1]:.3f}" for k, v in hist.items()), flush=True)
stop = True
print(f"\n{kind} maps, S={S} B={B} {layout} {mode}: median ms over [first 25 % of the run] / [last 50 %] / all   (frac of 8 TB/s from 'all')")
gb = 2 * B * S * S * 3 * 4 / 1e9
for k, v in hist.items():
    n = len(v)
    a = statistics.median(x for _, x in v[:max(1, n // 4)]); b = statistics.median(x for _, x in v[n // 2:]); c = statistics.median(x for _, x in v)
    print(f"{k:20s} {a:.4f} / {b:.4f} / {c:.4f}   {gb / c / 8:.3f}")
print("\nrocm-smi samples (every ~2 s):")
for t, d in smi_log[:: max(1, len(smi_log) // 12)]:
    print(f"t={t - t_start:6.1f}s {d}")
