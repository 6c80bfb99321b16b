#!/bin/bash
# round 5: rows per resample block / row blocks per workgroup of the one-launch chain step at B=256, 1024 -> 500
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
for t in "" "remap_rows:24" "remap_rows:32" "remap_rows:48" "remap_rows:64" "remap_cpw:2" "remap_rows:32,remap_cpw:2" "remap_rows:8,remap_cpw:4" ""; do
  python tools/attic/chain_stream_bench.py patterns=fused cases=256:1024:500,64:1024:500 tune=$t 2>&1 | grep -v amdgpu.ids
done
