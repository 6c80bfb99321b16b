"""Dev tool: same-box A/B of builds of libattwarp_hip.so (alternating subprocesses, one build per process).
usage: python tools/ab.py <target> libA.so libB.so [...]     targets: attn | remap | headline | step | u8
(variants inside ONE build are compared with attwarp_debug_set through the tools that take key=value / tune= arguments)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--child" in sys.argv:
    target = sys.argv[sys.argv.index("--child") + 1]
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    from attwarp_amd import _lib
    _lib.LIB_PATH = os.environ["AB_LIB"]
    tag = os.path.basename(os.environ["AB_LIB"])
    import torch
    from attwarp_amd import pipeline
    dev = torch.device("cuda:0")
    if target == "attn":
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
        print(tag, " ".join(res), flush=True)
    elif target == "remap":
        import remap_bench as rb
        for rep in range(2):
            rb.bench(256, 1024, "chw", "uniform", "cv2", tag=tag)
            rb.bench(256, 1024, "chw", "uniform", "exact", tag=tag)
            rb.bench(64, 336, "hwc", "uniform", "cv2", 100, tag=tag)
            rb.bench(256, 336, "hwc", "uniform", "cv2", 50, tag=tag)
            rb.bench(256, 336, "hwc", "uniform", "exact", 50, tag=tag)
            rb.bench(256, 336, "chw", "uniform", "cv2", 50, tag=tag)
        rb.bench(256, 512, "hwc", "uniform", "cv2", 50, tag=tag)
        rb.bench(256, 336, "hwc", "peaked", "cv2", 50, tag=tag)
        rb.bench(256, 1024, "chw", "peaked", "cv2", tag=tag)
        rb.bench(256, 1024, "hwc", "uniform", "cv2", tag=tag)
    elif target == "headline":
        import remap_bench as rb
        for rep in range(3):
            rb.bench(256, 1024, "hwc", "uniform", "cv2", tag=tag)
            rb.bench(256, 1024, "hwc", "uniform", "exact", tag=tag)
            rb.bench(256, 1024, "hwc", "peaked", "cv2", tag=tag)
    elif target == "step":
        import bench
        from attwarp_amd import dist as D
        for (B, S, K) in ((64, 336, 48), (256, 336, 24)):
            res, *_ = bench.small_workload(B, S, dev, 5, "cv2", "hwc", K, 2, D, torch, pipeline)
            print(f"{tag:12s} B={B} S={S}: fused {res['ms_per_step']:.4f} ms/step {res['step_TBps']:.3f} TB/s  same={res['bit_identical_to_serial']}  "
                  f"eager {res['eager']['ms_per_step']:.4f} stages {res['eager']['stages_ms']}", flush=True)
            torch.cuda.empty_cache()
    elif target == "u8":       # the integer uint8 resample alone and inside the one-launch chain steps (uniform + ragged)
        import remap_bench as rb, time
        for (B, S, So) in ((64, 336, 500), (256, 336, 500), (256, 1024, 500), (256, 1024, 1024), (256, 683, 500)):
            g = torch.Generator(device=dev).manual_seed(1)
            img = torch.randint(0, 256, (B, S if S != 683 else 1024, S, 3), device=dev, dtype=torch.uint8, generator=g)
            px = torch.softmax(torch.randn(B, 24, device=dev, generator=g) * 0.5, 1); py = torch.softmax(torch.randn(B, 24, device=dev, generator=g) * 0.5, 1)
            mx, my = pipeline.axis_maps_from_pdf(px, py, (img.shape[1], S), (So, So))
            out = torch.empty(B, So, So, 3, device=dev, dtype=torch.uint8)
            from attwarp_amd import checkpoint_utils as cu
            for _ in range(5): cu.remap_separable(img, mx, my, mode="cv2", channels_last=True, out=out)
            torch.cuda.synchronize(); ts = []
            for _ in range(40):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); cu.remap_separable(img, mx, my, mode="cv2", channels_last=True, out=out); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            print(f"{tag:22s} u8 cv2 B={B} W={S}->{So}: {sorted(ts)[20]*1e3:.1f} us", flush=True)
            del img, out
        for (B, S, K) in ((32, 336, 64), (64, 336, 64), (256, 336, 32), (64, 1024, 32), (256, 1024, 32)):
            n = 6 if S == 1024 or B == 256 else 16
            g = torch.Generator(device=dev).manual_seed(B + S)
            images = [torch.randint(0, 256, (B, S, S, 3), device=dev, dtype=torch.uint8, generator=g) for _ in range(n)]
            masks = [torch.rand(B, 24, 24, device=dev, generator=g) for _ in range(n)]
            mc = pipeline.MaskChainStream(images, masks, (500, 500), pattern="fused")
            def run():
                mc.reset(); mc.prime(); mc.run(K); mc.drain()
            run(); torch.cuda.synchronize(); best = 1e9
            for rep in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / (K + mc.depth))
            print(f"{tag:22s} chain step B={B} {S}->500: {best*1e6:.1f} us", flush=True)
            del mc, images, masks; torch.cuda.empty_cache()
        WH = [(1024, 768), (683, 1024), (1024, 1024), (500, 375), (333, 500), (640, 427)]
        for (B, K) in ((32, 60), (256, 24)):
            g = torch.Generator(device=dev).manual_seed(B)
            ring = []
            for _ in range(12 if B == 32 else 6):
                rb = pipeline.RaggedBatch([torch.randint(0, 256, (h, w, 3), device=dev, dtype=torch.uint8, generator=g)
                                           for (w, h) in (WH[b % 6] for b in range(B))], (500, 500))
                rb.masks = torch.rand(B, 24, 24, device=dev, generator=g)
                ring.append(rb)
            st = pipeline.RaggedMaskChainStream(out_size=(500, 500))
            st.ring(ring)
            def run_r():
                st.k = 0; st.prime(); st.run(K, unroll=len(ring)); st.drain_ring()
            run_r(); torch.cuda.synchronize(); best = 1e9
            for rep in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter(); run_r(); torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / (K + 4))
            print(f"{tag:22s} ragged step B={B}: {best*1e6:.1f} us", flush=True)
            del st, ring; torch.cuda.empty_cache()
    else:
        raise SystemExit(__doc__)
else:
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    target = sys.argv[1]
    libs = [os.path.abspath(p) for p in sys.argv[2:] if p.endswith(".so")]
    for rep in range(3 if target != "step" else 2):
        for lib in libs:
            subprocess.run([sys.executable, __file__, "--child", target], env=dict(os.environ, AB_LIB=lib))
