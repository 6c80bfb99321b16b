"""Dev tool: the main_batched chain (AGW/main_batched.py:243-287) as a batch STREAM -- pipeline.MaskChainStream patterns
against the serial launches of pipeline.warp_from_masks, over a ring of independent batches larger than the Infinity
Cache.  usage: python tools/chain_stream_bench.py [patterns=serial,branches,fused] [cases=32:336:500,...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from attwarp_amd import pipeline, _lib
dev = torch.device("cuda:0")
args = dict(a.split("=") for a in sys.argv[1:])
# tune=key:value,key:value  -> attwarp_debug_set overrides of the tuning flavour for the stream patterns (not the reference)
over = {k: int(v) for k, v in (kv.split(":") for kv in args.get("tune", "").split(",") if kv)}
patterns = args.get("patterns", "serial,branches,fused").split(",")
cases = [tuple(int(v) for v in c.split(":")) for c in args.get("cases", "32:336:500,64:336:500,256:336:500,32:1024:500,64:1024:500,256:1024:500").split(",")]
K = int(args.get("steps", "64"))

def chain_bytes(B, S, So):   # mask written + mask read by the marginals + image read + output written
    return B * (S * S + S * S + 3 * S * S + 3 * So * So)

for (B, S, So) in cases:
    slot = B * (3 * S * S + 3 * So * So)
    n = max(2, min(16, -(-(1 << 30) // slot)))
    n += n & 1
    g = torch.Generator(device=dev).manual_seed(B + S)
    images = [torch.randint(0, 256, (B, S, S, 3), device=dev, dtype=torch.uint8, generator=g) for _ in range(n)]
    masks = [torch.rand(B, 24, 24, device=dev, generator=g) for _ in range(n)]
    refs = [pipeline.warp_from_masks(images[i], masks[i], (So, So)) for i in range(min(n, 4))]
    for pat in patterns:
        import contextlib
        with (_lib.debug_override(**over) if over else contextlib.nullcontext()):
            try:
                mc = pipeline.MaskChainStream(images, masks, (So, So), pattern=pat)
            except Exception as e:
                print(f"B={B} {S}->{So} {pat}: unavailable ({e})", flush=True)
                continue
            def run(steps):
                mc.reset(); mc.prime(); mc.run(steps); mc.drain()
            run(K); torch.cuda.synchronize()      # (graphs are captured here, under the overrides)
        same = all(torch.equal(mc.outs[i], refs[i]) for i in range(len(refs)))
        best = 1e9
        for rep in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            run(K)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / (K + mc.depth))
        cb = chain_bytes(B, S, So)
        print(f"B={B:3d} {S:4d}->{So} {mc.pattern:9s} {args.get('tune', ''):28s} ring {n:2d}: {best*1e6:8.1f} us/step  {cb/best/1e12:6.3f} TB/s  "
              f"{cb/best/8e12:6.3f} of peak  bit_identical_to_serial={same}", flush=True)
        del mc
    del images, masks, refs
    torch.cuda.empty_cache()
