"""Dev tool: time the attention reduce (A1) alone at the bench shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attwarp_amd import pipeline
dev = torch.device("cuda:0")
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
for B in (256, 64):
    rows = torch.softmax(torch.randn(20, B, 32, 640, device=dev), -1)
    starts = (35 + torch.arange(B, device=dev) % 8).int()
    st = starts.repeat(20)
    for name, r in (("fp32", rows), ("fp16", rows.half()), ("bf16", rows.bfloat16())):
        for hu in (4, 1, 2, 8):
            from attwarp_amd import _lib
            with _lib.debug_override(attn_hu=hu):
                ms = timeit(lambda: pipeline.attention_step_maps(r, starts, 576, st))
            nb = 20 * B * 32 * 576 * r.element_size()
            print(f"attn step maps {name} T=20 B={B} HU={hu}: {ms:.4f} ms  {nb/ms/1e9:.2f} TB/s")
