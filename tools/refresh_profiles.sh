#!/bin/bash
# Run ON THE GPU BOX (gpurun -- bash tools/refresh_profiles.sh [tag]): regenerates the raw material of profiles/
# under gpurun_out/<tag>/.  tools/make_profiles.py turns it into the committed summaries.
# Order matters: the PMC traffic passes run FIRST and are summarised into profiles/pmc_traffic.json on the box, so that the
# bench line printed right behind them quotes the counters of ITS OWN lease (`roofline.traffic`, `traffic_source`).
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r6}
NAME=${2:-round6}
OUT=$ROOT/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"   # (on the GPU box; gpurun merges into the local gpurun_out/, where older files may remain)
cd /tmp && export TMPDIR=/tmp
keep() {  # keep only the summaries of a rocprofv3 output directory (traces are large)
  local d=$1
  mkdir -p $OUT/keep/$d
  find $OUT/$d -name "*kernel_stats.csv" -exec cp {} $OUT/keep/$d/ \;
  [ "$d" = trace ] && find $OUT/$d -name "*kernel_trace.csv" -exec cp {} $OUT/keep/$d/ \;
  find $OUT/$d -name "*counter_collection.csv" -exec cp {} $OUT/keep/$d/ \;
  rm -rf $OUT/$d
}
kstats() {  # kstats <name> <prof.py target...>: rocprofv3 --stats of one part of the path -> $OUT/<name>_kernel_stats.txt
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${name}_trace -- python3 $ROOT/tools/prof.py "$@" > /dev/null 2>&1
  python3 $ROOT/tools/kstats.py $(find $OUT/${name}_trace -name "*kernel_stats.csv" | head -1) > $OUT/${name}_kernel_stats.txt
  rm -rf $OUT/${name}_trace
}
# which lease this is: every number under $OUT -- the bench line AND the PMC traffic passes -- comes from this one box
{ echo "host $(hostname)  $(date -u +%Y-%m-%dT%H:%M:%SZ)"; rocm-smi --showuniqueid 2>/dev/null | grep -i "GPU\[" | head -1; } > $OUT/lease.txt

# 1. HBM traffic of the roofline kernel: separate --pmc passes (counters only), both arithmetic modes + the planar layout
for V in "cv2 hwc" "exact hwc" "cv2 chw"; do
  set -- $V; MODE=$1; LAY=$2; KEY=$MODE; [ $LAY = chw ] && KEY=${MODE}_chw
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$KEY -- python3 $ROOT/bench.py --no-cpu-baseline --no-also --steps 5 --mode $MODE --layout $LAY > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$KEY -- python3 $ROOT/bench.py --no-cpu-baseline --no-also --steps 5 --mode $MODE --layout $LAY > /dev/null 2>&1
  keep pmc_fetch_$KEY; keep pmc_write_$KEY
done
python3 $ROOT/tools/make_profiles.py $TAG $NAME --pmc-only > $OUT/pmc_traffic.log 2>&1     # -> profiles/pmc_traffic.json ON THIS BOX

# 2. the bench line (reads the pmc_traffic.json just written), then the same command under the kernel trace
python3 $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
# the headline agreement: ONLY the main measurement (25 pre-conditioning + 3 warm-up + 20 timed steps), so that the
# average duration of remap_rows_kernel in the stats is the average of the launches bench.py times
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-also > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
keep trace
# the whole default command (exact / CHW / fused / peaked / zero attention / 336 workloads share kernel names: its
# per-kernel averages mix those measurements)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_full -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/bench_profiled_full.json 2> $OUT/bench_profiled_full.err
keep trace_full

# 3. the other workloads as main lines; the line a rank of N prints (one-rank RCCL group, dist legs on "every" rank)
python3 $ROOT/bench.py --workload 336 --no-cpu-baseline > $OUT/bench_336.json 2> $OUT/bench_336.err
python3 $ROOT/bench.py --workload 336x256 --no-cpu-baseline > $OUT/bench_336x256.json 2> $OUT/bench_336x256.err
python3 $ROOT/bench.py --workload main_batched > $OUT/bench_main_batched.json 2> $OUT/bench_main_batched.err
python3 $ROOT/bench.py --workload main_batched_ragged > $OUT/bench_main_batched_ragged.json 2> $OUT/bench_main_batched_ragged.err
python3 $ROOT/bench.py --workload config5 > $OUT/bench_config5.json 2> $OUT/bench_config5.err
python3 $ROOT/bench.py --gpus 1 --force-dist --no-cpu-baseline --legs 336,main_batched_ragged > $OUT/bench_force_dist.json 2> $OUT/bench_force_dist.err

# 4. per-stage tools
for t in attn_bench u8_bench stage_bench chain_bench probe_bench remap_bench remap_lines; do
  python3 $ROOT/tools/$t.py 2>&1 | grep -v amdgpu.ids > $OUT/${t%_bench}_bench.txt
done
mv $OUT/remap_lines_bench.txt $OUT/remap_lines.txt
for c in "chain 32 336 500" "chain 256 1024 500" "step 64 336" "remap 256 1024" "ragged 32"; do
  python3 $ROOT/tools/gantt.py $c bin=8 2>&1 | grep -v amdgpu.ids > "$OUT/timeline_$(echo $c | tr ' ' _).txt"
done
kstats ragged ragged 32
kstats chain_step chain_step 256 1024 500
kstats chain chain 256 1024 500
cat $OUT/bench.json
