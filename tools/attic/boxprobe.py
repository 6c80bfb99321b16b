"""Dev tool: one line per lease -- device properties, the resample kernel (cv2, B=256 1024x1024 HWC) and torch.add."""
import os, sys, statistics, subprocess, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from attwarp_amd import checkpoint_utils as cu
from remap_bench import maps
dev = torch.device("cuda:0")
pr = torch.cuda.get_device_properties(0)
B, S = 256, 1024
img = torch.rand((B, S, S, 3), device=dev); out = torch.empty_like(img)
mx, my = maps(B, S, "uniform")
def t(fn, n=30):
    ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return statistics.median(ts[5:])
k1 = t(lambda: cu.remap_separable(img, mx, my, mode="cv2", channels_last=True, out=out))
ad = t(lambda: torch.add(img, 1.0, out=out))
k2 = t(lambda: cu.remap_separable(img, mx, my, mode="cv2", channels_last=True, out=out))
ex = t(lambda: cu.remap_separable(img, mx, my, mode="exact", channels_last=True, out=out))
smi = subprocess.run("rocm-smi --showuniqueid --showserial --showbus 2>/dev/null | grep -i 'unique\\|serial\\|pci' | head -3 | tr '\\n' ' '", shell=True, capture_output=True, text=True).stdout
host = subprocess.run("hostname; nproc; cat /proc/cpuinfo | grep 'model name' | head -1", shell=True, capture_output=True, text=True).stdout.replace("\n", " ")
print(f"BOX {pr.name} CUs={pr.multi_processor_count} clk={getattr(pr,'clock_rate',None)} memclk={getattr(pr,'memory_clock_rate',None)} L2={getattr(pr,'L2_cache_size',None)} | remap cv2 {k1:.4f} / {k2:.4f} ms exact {ex:.4f} ms add {ad:.4f} ms | ratio {k2/ad:.3f} | {smi} | {host}")
