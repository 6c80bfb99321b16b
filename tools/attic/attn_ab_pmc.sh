#!/bin/bash
# Dev tool: same-box A/B of attention-reduce builds (timing) + their SQ instruction counters.
# usage: bash tools/attn_ab_pmc.sh outdir libA.so libB.so ...
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT" || exit 1
out=$1; shift
mkdir -p "$out"
python tools/ab.py attn "$@" > "$out/ab.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  n=$(basename "$lib" .so)
  # (the program follows `--` directly: no wrapper may exec between the profiler and python)
  if ! AB_LIB=$ROOT/$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace \
       -d "$ROOT/$out/pmc_$n" --output-format csv -- python3 "$ROOT/tools/prof.py" attn > "$ROOT/$out/pmc_$n.log" 2>&1; then
    echo "== $n: the rocprofv3 pass FAILED, see $out/pmc_$n.log" >> "$ROOT/$out/ab.txt"
    continue
  fi
  f=$(find "$ROOT/$out/pmc_$n" -name "*counter_collection.csv" | head -1)
  echo "== $n" >> "$ROOT/$out/ab.txt"
  python3 "$ROOT/tools/pmc_summary.py" "$f" attn_reduce >> "$ROOT/$out/ab.txt" 2>&1
done
cat "$ROOT/$out/ab.txt"
