#!/bin/bash
# Run ON THE GPU BOX (gpurun -- bash tools/refresh_profiles.sh): regenerates the raw material of profiles/
# under gpurun_out/r1/.  tools/make_profiles.py turns it into the committed summaries.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r1
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --no-cpu-baseline --no-also --steps 5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --no-cpu-baseline --no-also --steps 5 > /dev/null 2>&1
python3 $ROOT/tools/stage_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/stage_bench.txt
python3 $ROOT/tools/probe_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/probe_bench.txt
cat $OUT/bench.json
