"""Dev tool: A/B two builds of libattwarp_hip.so on the same box (alternating subprocesses).
usage: ab_attn.py libA.so libB.so   (child mode: AB_LIB=path ab_attn.py --child)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--child" in sys.argv:
    sys.path.insert(0, ROOT)
    from attwarp_amd import _lib
    _lib.LIB_PATH = os.environ["AB_LIB"]
    import torch
    from attwarp_amd import pipeline
    dev = torch.device("cuda:0")
    B = 256
    rows = torch.softmax(torch.randn(20, B, 32, 640, device=dev), -1)
    starts = (35 + torch.arange(B, device=dev) % 8).int()
    st = starts.repeat(20)
    res = []
    for name, r in (("fp32", rows), ("fp16", rows.half()), ("bf16", rows.bfloat16())):
        for _ in range(5): pipeline.attention_step_maps(r, starts, 576, st)
        torch.cuda.synchronize(); ts = []
        for _ in range(40):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); pipeline.attention_step_maps(r, starts, 576, st); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res.append(f"{name} {sorted(ts)[len(ts)//2]*1e3:.1f}us")
    print(os.path.basename(os.environ["AB_LIB"]), " ".join(res), flush=True)
else:
    libs = [os.path.abspath(p) for p in sys.argv[1:3]]
    for rep in range(3):
        for lib in libs:
            subprocess.run([sys.executable, __file__, "--child"], env=dict(os.environ, AB_LIB=lib))
