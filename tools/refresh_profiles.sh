#!/bin/bash
# Run ON THE GPU BOX (gpurun -- bash tools/refresh_profiles.sh [tag]): regenerates the raw material of profiles/
# under gpurun_out/<tag>/.  tools/make_profiles.py turns it into the committed summaries.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r5}
OUT=$ROOT/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"   # (on the GPU box; gpurun merges into the local gpurun_out/, where older files may remain)
cd /tmp && export TMPDIR=/tmp
# which lease this is: every number under $OUT -- the bench line AND the PMC traffic passes -- comes from this one box
{ echo "host $(hostname)  $(date -u +%Y-%m-%dT%H:%M:%SZ)"; rocm-smi --showuniqueid 2>/dev/null | grep -i "GPU\[" | head -1; } > $OUT/lease.txt
python3 $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
# the headline agreement: ONLY the main measurement (25 pre-conditioning + 3 warm-up + 20 timed steps), so that the
# average duration of remap_rows_kernel in the stats is the average of the launches bench.py times
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-also > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
# the whole default command (exact / CHW / fused / peaked / zero attention / 336 workloads share kernel names: its
# per-kernel averages mix those measurements)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_full -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/bench_profiled_full.json 2> $OUT/bench_profiled_full.err
for MODE in cv2 exact; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$MODE -- python3 $ROOT/bench.py --no-cpu-baseline --no-also --steps 5 --mode $MODE > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$MODE -- python3 $ROOT/bench.py --no-cpu-baseline --no-also --steps 5 --mode $MODE > /dev/null 2>&1
done
# the planar layout of the also_chw line (key 1024_cv2_chw of profiles/pmc_traffic.json)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_cv2_chw -- python3 $ROOT/bench.py --no-cpu-baseline --no-also --steps 5 --mode cv2 --layout chw > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_cv2_chw -- python3 $ROOT/bench.py --no-cpu-baseline --no-also --steps 5 --mode cv2 --layout chw > /dev/null 2>&1
python3 $ROOT/bench.py --workload 336 --no-cpu-baseline > $OUT/bench_336.json 2> $OUT/bench_336.err
python3 $ROOT/bench.py --workload 336x256 --no-cpu-baseline > $OUT/bench_336x256.json 2> $OUT/bench_336x256.err
python3 $ROOT/tools/attn_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/attn_bench.txt
python3 $ROOT/tools/u8_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/u8_bench.txt
python3 $ROOT/tools/stage_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/stage_bench.txt
python3 $ROOT/tools/chain_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/chain_bench.txt
python3 $ROOT/tools/probe_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/probe_bench.txt
python3 $ROOT/tools/remap_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/remap_bench.txt
python3 $ROOT/tools/chain_stream_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/chain_stream.txt
python3 $ROOT/tools/pair_step_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/pair_step.txt
for c in "chain 32 336 500" "chain 64 336 500" "chain 256 1024 500" "step 64 336" "step 256 336" "remap 256 1024" "remap 256 336" "ragged 32" "ragged 256"; do
  python3 $ROOT/tools/gantt.py $c bin=8 2>&1 | grep -v amdgpu.ids > "$OUT/timeline_$(echo $c | tr ' ' _).txt"
done
python3 $ROOT/tools/remap_lines.py 2>&1 | grep -v amdgpu.ids > $OUT/remap_lines.txt
python3 $ROOT/bench.py --workload main_batched > $OUT/bench_main_batched.json 2> $OUT/bench_main_batched.err
python3 $ROOT/bench.py --workload main_batched_ragged > $OUT/bench_main_batched_ragged.json 2> $OUT/bench_main_batched_ragged.err
bash $ROOT/tools/bounds.sh 2>&1 | grep -v amdgpu.ids > $OUT/bounds.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ragged_trace -- python3 $ROOT/tools/prof.py ragged 32 > /dev/null 2>&1
python3 $ROOT/tools/kstats.py $(find $OUT/ragged_trace -name "*kernel_stats.csv" | head -1) > $OUT/ragged_kernel_stats.txt
rm -rf $OUT/ragged_trace
cd /tmp
python3 $ROOT/bench.py --workload config5 > $OUT/bench_config5.json 2> $OUT/bench_config5.err
python3 $ROOT/bench.py --gpus 1 --force-dist --no-cpu-baseline --legs none > $OUT/bench_force_dist.json 2> $OUT/bench_force_dist.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/chain_step_trace -- python3 $ROOT/tools/prof.py chain_step 256 1024 500 > /dev/null 2>&1
python3 $ROOT/tools/kstats.py $(find $OUT/chain_step_trace -name "*kernel_stats.csv" | head -1) > $OUT/chain_step_kernel_stats.txt
rm -rf $OUT/chain_step_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/chain_trace -- python3 $ROOT/tools/prof.py chain 256 1024 500 > /dev/null 2>&1
python3 $ROOT/tools/kstats.py $(find $OUT/chain_trace -name "*kernel_stats.csv" | head -1) > $OUT/chain_kernel_stats.txt
# keep only the summaries (traces are large)
for d in trace trace_full pmc_fetch_cv2 pmc_write_cv2 pmc_fetch_exact pmc_write_exact pmc_fetch_cv2_chw pmc_write_cv2_chw; do
  mkdir -p $OUT/keep/$d
  find $OUT/$d -name "*kernel_stats.csv" -exec cp {} $OUT/keep/$d/ \;
  [ "$d" = trace ] && find $OUT/$d -name "*kernel_trace.csv" -exec cp {} $OUT/keep/$d/ \;
  find $OUT/$d -name "*counter_collection.csv" -exec cp {} $OUT/keep/$d/ \;
  rm -rf $OUT/$d
done
rm -rf $OUT/chain_trace
cat $OUT/bench.json
