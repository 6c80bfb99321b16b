#!/bin/bash
# Dev tool: block order x occupancy x resample rows-per-block of the one-launch mask-chain step (one process per variant).
out=${1:-gpurun_out/r4/chain_sweep.txt}
: > $out
cases=${CASES:-256:336:500,64:1024:500,256:1024:500}
python tools/attic/chain_stream_bench.py patterns=branches cases=$cases >> $out 2>&1
for seq in 0 1 2 3 4; do for waves in 6 8; do for rows in -1 16 8; do
  t="chain_seq:$seq,chain_waves:$waves"; [ $rows -gt 0 ] && t="$t,remap_rows:$rows"
  python tools/attic/chain_stream_bench.py patterns=fused cases=$cases tune=$t 2>&1 | grep "us/step" >> $out
done; done; done
cat $out
