#!/bin/bash
# Upper-bound experiments of round 5 (tuning flavour, key "bound": outputs are garbage, only the time counts) -- what would
# the ideas of VERDICT r4 items 6-8 save AT MOST?  One process per line, each alternating baseline / bound.
#   bit 0 (1): float32 cv2 resample stages only the top row in a 3-row LDS pool      -> a three-slot LDS ring's ceiling
#   bit 1 (2): chain step, every image's up-sampled mask aliases image 0 / 1          -> fusing LANCZOS into the marginals body
#   bit 2 (4): finalize body without its np.cumsum chain;  bit 3 (8): finalize body returns at once
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
python - <<'PY'
import os, sys, io, contextlib
sys.path.insert(0, "tools")
import remap_bench as rb
res = {0: [], 1: []}
for rep in range(4):
    for b in (0, 1):
        with contextlib.redirect_stdout(io.StringIO()):
            res[b].append(rb.bench(256, 1024, "hwc", "uniform", "cv2", 10, **({"bound": 1} if b else {})))
print("E1 headline cv2 resample 256x1024x1024x3: baseline ms", [round(v, 4) for v in res[0]], " top-row-only, 3-row LDS pool ms", [round(v, 4) for v in res[1]], flush=True)
PY
for t in "" "bound:2"; do python tools/attic/chain_stream_bench.py patterns=fused cases=256:1024:500 tune=$t; done
for t in "" "bound:4" "bound:8" ""; do python tools/attic/chain_stream_bench.py patterns=fused cases=32:336:500,64:336:500 tune=$t; done
