#!/bin/bash
# round 5: rows per resample block / block order of the one-launch chain step at the reference's scale (B=32/64, 336 -> 500)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
for t in "" "remap_rows:8" "remap_rows:10" "remap_rows:12" "remap_rows:24" "remap_rows:32" "chain_seq:0" "chain_seq:1" "chain_waves:6" "remap_rows:8,chain_seq:0" ""; do
  python tools/attic/chain_stream_bench.py patterns=fused cases=32:336:500,64:336:500 tune=$t 2>&1 | grep -v amdgpu.ids
done
