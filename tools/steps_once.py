"""Dev tool: axis_maps_from_steps launches at the bench shapes for rocprofv3 --kernel-trace (per-dispatch durations
grouped by grid size): python tools/steps_once.py  /  python tools/steps_once.py summarize <kernel_trace.csv>"""
import csv, os, statistics, sys
if len(sys.argv) > 2 and sys.argv[1] == "summarize":
    by = {}
    for r in csv.DictReader(open(sys.argv[2])):
        n = r["Kernel_Name"]
        if "axis_maps_from_steps" not in n and "attn_reduce_step" not in n:
            continue
        key = (n.split("(")[0][-60:], r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Grid_Size_Y"))
        by.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in sorted(by.items()):
        print(f"{k}: n={len(v)} median {statistics.median(v):.2f} us  min {min(v):.2f}  max {max(v):.2f}")
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attwarp_amd import pipeline
dev = torch.device("cuda:0")
for (B, S) in [(64, 336), (256, 336), (256, 1024), (8, 336)]:
    rows = torch.softmax(torch.randn(20, B, 32, 640, device=dev), -1)
    starts = (35 + torch.arange(B, device=dev) % 8).to(torch.int32)
    for _ in range(12):
        steps = pipeline.attention_step_maps(rows, starts)
        pipeline.axis_maps_from_attention_steps(steps, (S, S))
    torch.cuda.synchronize()
