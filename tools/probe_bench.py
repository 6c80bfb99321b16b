"""Dev tool: time the last-query attention probe against what the reference's hook needs:
eager attention with output_attentions on the target layer (QK^T for every query row + softmax)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attwarp_amd import attention_extraction as ae
dev = torch.device("cuda:0")

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]

for (B, H, Hkv, kv, D, dt) in [(64, 32, 32, 640, 128, torch.float16), (256, 32, 32, 640, 128, torch.float16),
                                (64, 32, 32, 660, 128, torch.float16), (64, 32, 8, 640, 128, torch.bfloat16),
                                (64, 32, 32, 640, 128, torch.float32)]:
    q = torch.randn(B, H, D, device=dev, dtype=dt)
    k = torch.randn(B, Hkv, kv, D, device=dev, dtype=dt)
    starts = (35 + torch.arange(B, device=dev) % 8).int()
    pads = (torch.arange(B, device=dev) % 8).int()
    ms = timeit(lambda: ae.probe_last_query(q, k, starts, 576, pads))
    nbytes = B * Hkv * kv * D * q.element_size()
    print(f"probe B={B} H={H}/{Hkv} kv={kv} D={D} {str(dt)[6:]:8s}: {ms:.4f} ms  K bytes {nbytes/1e6:.0f} MB  "
          f"{nbytes/ms/1e9:.2f} TB/s")
    if B == 64 and Hkv == H and dt == torch.float16:
        # the reference's way at prefill: every query row (q = kv) of one layer, eager
        qf = torch.randn(B, H, kv, D, device=dev, dtype=dt)
        def eager():
            w = torch.matmul(qf, k.transpose(2, 3)) * (D ** -0.5)
            p = torch.softmax(w, dim=-1, dtype=torch.float32).to(dt)
            return ae.attn_reduce_step(p, starts, 576)
        print(f"   eager prefill weights [B,H,{kv},{kv}] + softmax + hook reduce: {timeit(eager, 5):.3f} ms")
        q1 = qf[:, :, -1:]
        def eager_decode():
            w = torch.matmul(q1, k.transpose(2, 3)) * (D ** -0.5)
            p = torch.softmax(w, dim=-1, dtype=torch.float32).to(dt)
            return ae.attn_reduce_step(p, starts, 576)
        print(f"   eager decode  weights [B,H,1,{kv}] + softmax + hook reduce: {timeit(eager_decode, 10):.3f} ms")
