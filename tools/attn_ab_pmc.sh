#!/bin/bash
# Dev tool: same-box A/B of attention-reduce builds (timing) + their SQ instruction counters.
# usage: bash tools/attn_ab_pmc.sh outdir libA.so libB.so ...
out=$1; shift
mkdir -p $out
python tools/ab.py attn "$@" > $out/ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  n=$(basename $lib .so)
  AB_LIB=$GRAFT_REPO_ROOT/$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace -d $GRAFT_REPO_ROOT/$out/pmc_$n --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/prof.py attn > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/$out/pmc_$n -name "*counter_collection.csv" | head -1)
  echo "== $n" >> $GRAFT_REPO_ROOT/$out/ab.txt
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $f attn_reduce >> $GRAFT_REPO_ROOT/$out/ab.txt 2>&1
done
cat $GRAFT_REPO_ROOT/$out/ab.txt
