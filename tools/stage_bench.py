"""Dev tool: time every stage kernel at BASELINE sizes (ms, algorithmic GB/s)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attwarp_amd import checkpoint_utils as cu, pipeline, attention_extraction as ae, new_method as nm, model, _lib
from attwarp_amd._lib import call, ptr, stream_ptr
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
def rep(name, ms, nbytes): print(f"{name:58s} {ms:9.4f} ms  {nbytes/ms/1e6:9.1f} GB/s")
for (B, S) in [(256, 1024), (64, 336)]:
    A = torch.rand(B, 1, S, S, device=dev)
    out = torch.empty(B, 24, 24, device=dev)
    a3 = A[:, 0].contiguous()
    rep(f"pool24 fp32 B={B} S={S}", timeit(lambda: call("attwarp_adaptive_avg_pool", ptr(a3), B, S, S, 24, 24, 0, ptr(out), stream_ptr(dev))), B*S*S*4)
    rep(f"gt_marginals full-res fp32 B={B} S={S}", timeit(lambda: cu.gt_marginals(A)), B*S*S*4)
    au8 = (A[:, 0] * 255).to(torch.uint8).contiguous()
    rep(f"attention_axis_maps u8 identity B={B} S={S}", timeit(lambda: nm.attention_axis_maps(au8, 500, 500, "identity")), B*S*S)
    rep(f"attention_axis_maps f32 identity B={B} S={S}", timeit(lambda: nm.attention_axis_maps(a3, S, S, "identity")), B*S*S*4)
    rep(f"attention_axis_maps f32 sqrt B={B} S={S}", timeit(lambda: nm.attention_axis_maps(a3, S, S, "sqrt")), B*S*S*4)
    from attwarp_amd import _lib
    with _lib.debug_override(profiles_variant=2):
        rep(f"attention_axis_maps f32 sqrt (variant 2) B={B} S={S}", timeit(lambda: nm.attention_axis_maps(a3, S, S, "sqrt")), B*S*S*4)
    with _lib.debug_override(profiles_variant=1):
        rep(f"gt_marginals full-res fp32 (variant 1) B={B} S={S}", timeit(lambda: cu.gt_marginals(A)), B*S*S*4)
        rep(f"attention_axis_maps f32 identity (variant 1) B={B} S={S}", timeit(lambda: nm.attention_axis_maps(a3, S, S, "identity")), B*S*S*4)
    m24 = torch.rand(B, 24, 24, device=dev)
    rep(f"revise_mask B={B}", timeit(lambda: ae.revise_mask(m24)), B*576*8)
    rep(f"upsample_mask_lanczos 24->{S} B={B}", timeit(lambda: ae.upsample_mask_lanczos(m24, (S, S))), B*S*S)
    rows = torch.softmax(torch.randn(20, B, 32, 640, device=dev), -1)
    starts = (35 + torch.arange(B, device=dev) % 8).int()
    st = starts.repeat(20)
    rep(f"attn step maps fp32 T=20 B={B}", timeit(lambda: pipeline.attention_step_maps(rows, starts, 576, st)), 20*B*32*576*4)
    r16 = rows.half()
    rep(f"attn step maps fp16 T=20 B={B}", timeit(lambda: pipeline.attention_step_maps(r16, starts, 576, st)), 20*B*32*576*2)
    steps = pipeline.attention_step_maps(rows, starts, 576, st)
    rep(f"fused maps from steps B={B} S={S}", timeit(lambda: pipeline.axis_maps_from_attention_steps(steps, (S, S))), 20*B*576*4)
    px = torch.softmax(torch.randn(B, 24, device=dev), 1)
    rep(f"axis_maps_from_pdf B={B} S={S}", timeit(lambda: pipeline.axis_maps_from_pdf(px, px, (S, S))), B*2*S*4)
    F = cu.cdf_from_density(cu.upsample_pdf_right_inverse(px, S).clamp_min(0))
    rep(f"axis_maps_from_cdf (2 launches) B={B} S={S}", timeit(lambda: cu.axis_maps_from_cdf(F, F, (S, S))), B*2*S*8)
    img8 = (torch.rand(B, S, S, 3, device=dev) * 255).to(torch.uint8)
    mx, my = pipeline.axis_maps_from_pdf(px, px, (S, S), (500, 500))
    rep(f"remap u8 HWC {S}->500 B={B}", timeit(lambda: nm.remap_hwc(img8, mx, my)), B*(S*S*3 + 500*500*3))
    if S == 336:
        w500 = (torch.rand(B, 500, 500, 3, device=dev) * 255).to(torch.uint8)
        rep(f"clip_preprocess 500->336 u8 -> f16 [B,3,336,336] B={B}", timeit(lambda: pipeline.clip_preprocess(w500)), B*(500*500*3 + 3*336*336*2))
        rep(f"pipeline.warp_from_masks (main_batched chain) B={B}", timeit(lambda: pipeline.warp_from_masks(img8, m24)), B*(S*S*3 + 500*500*3))
    del A, a3, au8, rows, r16, img8
# "next" row 1: MarginalNet(1024, 4096, 256) tail at config-5 shapes, fused kernels vs the stock ops of the reference
B = 256
v = torch.randn(B, 256, 24, 24, device=dev); gb = torch.randn(B, 512, device=dev)
tok = torch.randn(B, 32, 4096, device=dev); msk = (torch.rand(B, 32, 1, device=dev) > 0.3).float()
def stock_film():
    g, b_ = gb.chunk(2, dim=1)
    w = g[:, :, None, None] * v + b_[:, :, None, None]
    return w.mean(dim=2), w.mean(dim=3)
def stock_tmean():
    return (tok * msk).sum(dim=1) / msk.sum(dim=1).clamp_min(1.0)
rep(f"film_axis_means fused B={B} Ch=256 24x24", timeit(lambda: model.film_axis_means(v, gb)), v.numel() * 4)
rep(f"film + 2 means, stock torch ops B={B}", timeit(stock_film), v.numel() * 4)
rep(f"masked_token_mean fused B={B} Lt=32 D=4096", timeit(lambda: model.masked_token_mean(tok, msk)), tok.numel() * 4)
rep(f"masked mean, stock torch ops B={B}", timeit(stock_tmean), tok.numel() * 4)
net = model.MarginalNet(1024, 4096, 256).to(dev).eval()
fmap = torch.randn(B, 1024, 24, 24, device=dev)
with torch.no_grad():
    rep(f"MarginalNet forward (GEMMs stock + fused tail) B={B}", timeit(lambda: net(fmap, 24, 24, tok, msk), 5), fmap.numel() * 4)
    def stock_fwd():
        lx, ly = net.forward_logits(fmap, 24, 24, tok, msk)
        return model.safe_softmax(lx), model.safe_softmax(ly)
    rep(f"MarginalNet forward (all stock ops + safe_softmax) B={B}", timeit(stock_fwd, 5), fmap.numel() * 4)
