#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 PMC passes (counters only, no tracing domains besides --kernel-trace) over one tool
# script, summarised per kernel.   usage: bash tools/pmc.sh <tag> <tool.py> [args...]   -> gpurun_out/pmc_<tag>.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$ROOT/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
P4="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
P5="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum"
P6="TCC_EA0_RDREQ_GMI_32B_sum TCC_EA0_RDREQ_IO_32B_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"
P7="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_64B_sum"
i=0
for P in "$P1" "$P2" "$P3" "FETCH_SIZE" "WRITE_SIZE" "$P4" "$P5" "$P6" "$P7"; do
  i=$((i+1))
  [ $i -gt ${PMC_PASSES:-9} ] && break      # PMC_PASSES=5: the SQ passes + FETCH_SIZE / WRITE_SIZE only
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/"$@" > $OUT/p$i.log 2>&1 || echo "pass $i ($P) failed" >> $OUT/errors.txt
done
{
  echo "# rocprofv3 --pmc <counters> --kernel-trace -- python3 tools/$*   (9 separate passes; SQ cycle counters are quad-cycles;"
  echo "# FETCH_SIZE / WRITE_SIZE in KB: FETCH_SIZE x2 on gfx950 per MI355X_MICROARCH.md)"
  for d in $OUT/p*/; do
    f=$(find $d -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 $ROOT/tools/pmc_summary.py "$f"
  done
  [ -f $OUT/errors.txt ] && cat $OUT/errors.txt
} > $ROOT/gpurun_out/pmc_$TAG.txt 2>&1
rm -rf $OUT/p*/
tail -80 $ROOT/gpurun_out/pmc_$TAG.txt
