"""Dev tool: the main_batched chain (and chain + clip_preprocess) eager vs HIP-graph replay (pipeline.capture_step)."""
import os, sys, torch
sys.path.insert(0, "/root/repo")
from attwarp_amd import pipeline
dev = torch.device("cuda:0")
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
for B in (64, 8, 1):
    img8 = (torch.rand(B, 336, 336, 3, device=dev) * 255).to(torch.uint8)
    m24 = torch.rand(B, 24, 24, device=dev)
    eager = t(lambda: pipeline.warp_from_masks(img8, m24))
    ref = pipeline.warp_from_masks(img8, m24).clone()
    g, out = pipeline.capture_step(pipeline.warp_from_masks, img8, m24)
    rep = t(lambda: g.replay())
    print(f"B={B}: eager {eager*1e3:.1f} us, graph replay {rep*1e3:.1f} us, equal={torch.equal(out, ref)}")
    def chain():
        w = pipeline.warp_from_masks(img8, m24)
        return pipeline.clip_preprocess(w)
    eager2 = t(chain)
    g2, out2 = pipeline.capture_step(chain)
    rep2 = t(lambda: g2.replay())
    print(f"B={B}: masks->warp->clip eager {eager2*1e3:.1f} us, graph replay {rep2*1e3:.1f} us")
