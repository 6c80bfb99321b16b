"""Test infrastructure: concurrency stress.  N host threads, each with its own HIP stream, run random stage calls
(resample, maps from attention, PDF chain, LANCZOS up-sample, attention reduce, CLIP epilogue, ragged chain batches) on private inputs and
compare every result with the one the same call produced serially beforehand: kernels are stateless and stream ordered,
the host-side table caches are shared.   usage: fuzz_streams.py [seconds] [threads] [seed]"""
import os, sys, threading, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from attwarp_amd import checkpoint_utils as cu, attention_extraction as ae, new_method as nm, pipeline
dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0

def make_jobs(rng, n):
    jobs = []
    for _ in range(n):
        k = int(rng.integers(0, 7))
        if k == 0:
            B, H, W, C = int(rng.integers(1, 4)), int(rng.integers(8, 400)), int(rng.integers(8, 400)), int(rng.integers(1, 5))
            Ho, Wo = int(rng.integers(4, 400)), int(rng.integers(4, 400))
            img = torch.rand(B, H, W, C, device=dev) if rng.random() < 0.5 else (torch.rand(B, H, W, C, device=dev) * 255).to(torch.uint8)
            mx = (torch.rand(B, Wo, device=dev) * W).sort(1)[0]; my = (torch.rand(B, Ho, device=dev) * H).sort(1)[0]
            mode = "cv2" if rng.random() < 0.5 else "exact"
            jobs.append(lambda img=img, mx=mx, my=my, mode=mode: cu.remap_separable(img, mx, my, mode=mode, channels_last=True))
        elif k == 1:
            B, h, w = int(rng.integers(1, 4)), int(rng.integers(8, 500)), int(rng.integers(8, 500))
            att = (torch.rand(B, h, w, device=dev) * 255).to(torch.uint8)
            tr = str(rng.choice(["identity", "sqrt", "log"]))
            jobs.append(lambda att=att, tr=tr: torch.cat(nm.attention_axis_maps(att, 300, 200, tr), 1))
        elif k == 2:
            B, W, H = int(rng.integers(1, 5)), int(rng.integers(24, 900)), int(rng.integers(24, 900))
            px = torch.softmax(torch.randn(B, 24, device=dev), 1); py = torch.softmax(torch.randn(B, 24, device=dev), 1)
            jobs.append(lambda px=px, py=py, W=W, H=H: torch.cat(pipeline.axis_maps_from_pdf(px, py, (H, W)), 1))
        elif k == 3:
            B, W, H = int(rng.integers(1, 4)), int(rng.integers(24, 700)), int(rng.integers(24, 700))
            m = torch.rand(B, 24, 24, device=dev)
            jobs.append(lambda m=m, W=W, H=H: ae.upsample_mask_lanczos(ae.revise_mask(m), (W, H)))
        elif k == 4:
            T_, B, heads, kv = int(rng.integers(1, 5)), int(rng.integers(1, 5)), int(rng.integers(1, 33)), 576 + int(rng.integers(0, 64))
            rows = torch.softmax(torch.randn(T_, B, heads, kv, device=dev), -1)
            starts = torch.randint(0, kv - 576 + 1, (B,), device=dev, dtype=torch.int32)
            jobs.append(lambda rows=rows, starts=starts: ae.attn_reduce_stack(rows, starts, 576))
        elif k == 5:
            B, H, W = int(rng.integers(1, 3)), int(rng.integers(16, 600)), int(rng.integers(16, 600))
            img = (torch.rand(B, H, W, 3, device=dev) * 255).to(torch.uint8)
            jobs.append(lambda img=img: pipeline.clip_preprocess(img, 224, torch.float32))
        else:       # a ragged batch: host-side plan, the shared staging pool and table caches, five ragged launches
            B = int(rng.integers(1, 5))
            imgs = [(torch.rand(int(rng.integers(25, 300)), int(rng.integers(25, 300)), 3, device=dev) * 255).to(torch.uint8) for _ in range(B)]
            m = torch.rand(B, 24, 24, device=dev)
            jobs.append(lambda imgs=imgs, m=m: pipeline.warp_from_masks_ragged(imgs, m, (60, 72)))
    return jobs

rng = np.random.default_rng(seed)
jobs = [make_jobs(rng, 60) for _ in range(nthreads)]
expect = [[j().clone() for j in js] for js in jobs]            # serial pass on the default stream
torch.cuda.synchronize()
bad = [0] * nthreads; done = [0] * nthreads
t_end = time.time() + budget
def worker(i):
    st = torch.cuda.Stream(device=dev)
    r = np.random.default_rng(seed * 100 + i)
    with torch.cuda.stream(st):
        while time.time() < t_end:
            order = r.permutation(len(jobs[i]))
            outs = [(k, jobs[i][k]()) for k in order]
            st.synchronize()
            for k, o in outs:
                done[i] += 1
                if not torch.equal(o, expect[i][k]) and not (o.is_floating_point() and torch.equal(torch.nan_to_num(o), torch.nan_to_num(expect[i][k]))):
                    bad[i] += 1
ths = [threading.Thread(target=worker, args=(i,)) for i in range(nthreads)]
[t.start() for t in ths]; [t.join() for t in ths]
print(f"{sum(done)} calls on {nthreads} threads / streams, {sum(bad)} mismatches against the serial results")
sys.exit(1 if sum(bad) else 0)
