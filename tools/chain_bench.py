"""Dev tool: the main_batched chain (AGW/main_batched.py:243-287) stage by stage at B=256, 1024^2 -> 500^2 and
B=64, 336^2 -> 500^2: revise_mask, Lanczos mask up-sample, A13 maps from the uint8 mask, uint8 resample, whole chain."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attwarp_amd import pipeline, attention_extraction as ae, new_method as nm, _lib
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
def rep(name, ms, nbytes): print(f"{name:64s} {ms*1e3:9.1f} us  {nbytes/ms/1e6:9.1f} GB/s  {nbytes/ms/1e6/80:5.1f}% of 8 TB/s", flush=True)
over = {k: int(v) for k, v in (a.split("=") for a in sys.argv[1:])}
with _lib.debug_override(**over):
    for (B, S, So) in [(256, 1024, 500), (256, 1024, 1024), (64, 336, 500), (256, 336, 500)]:
        m24 = torch.rand(B, 24, 24, device=dev)
        img8 = (torch.rand(B, S, S, 3, device=dev) * 255).to(torch.uint8)
        rev = ae.revise_mask(m24)
        rep(f"revise_mask B={B}", timeit(lambda: ae.revise_mask(m24)), B * 576 * 8)
        rep(f"upsample_mask_lanczos 24->{S} B={B}", timeit(lambda: ae.upsample_mask_lanczos(rev, (S, S))), B * S * S)
        mota = ae.upsample_mask_lanczos(rev, (S, S))
        for tr in ("identity", "sqrt"):
            rep(f"attention_axis_maps u8 {tr} {S}->{So} B={B}", timeit(lambda: nm.attention_axis_maps(mota, So, So, tr)), B * S * S)
        mx, my = nm.attention_axis_maps(mota, So, So, "identity")
        for mode in ("cv2", "exact"):
            rep(f"remap u8 HWC {mode} {S}->{So} B={B}", timeit(lambda: nm.remap_hwc(img8, mx, my, mode)), B * (S * S * 3 + So * So * 3))
        rep(f"warp_from_masks (whole chain) {S}->{So} B={B}", timeit(lambda: pipeline.warp_from_masks(img8, m24, (So, So))), B * (S * S * 3 + So * So * 3))
        del img8, mota
