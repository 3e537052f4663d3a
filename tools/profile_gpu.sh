#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats pass + separate PMC passes of a short bench run.
# Usage: tools/profile_gpu.sh <tag>      outputs under gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 5 --warmup 2 --no-cpu --no-tiles --ff-rows 4096"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -5 $OUT/trace.log; exit 1; }
echo "trace pass done"
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc$i -o pmc -- $BENCH > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i ($C) failed"; tail -3 $OUT/pmc$i.log; exit 1; }
  echo "pmc pass $i done: $C"
done
find $OUT -name "*.csv" | head -30
