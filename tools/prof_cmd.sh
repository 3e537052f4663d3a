#!/bin/bash
# Runs on the GPU box (through gpurun): a kernel-trace pass + SQ / TCC PMC passes of ONE short python tool run, then
# prints per-kernel averages.   Usage: tools/prof_cmd.sh <tag> <kernel-name-substring> <script.py> [args...]
# outputs under gpurun_out/prof_<tag>/   (each rocprofv3 run gets the program itself after "--", never a shell)
set -u
TAG=$1; KSUB=$2; shift 2
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 "$@" > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -n 5 $OUT/trace.log; exit 1; }
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" ${EXTRA_PMC:+"$EXTRA_PMC"}; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc$i -o pmc -- python3 "$@" > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i ($C) failed"; tail -n 3 $OUT/pmc$i.log; exit 1; }
done
python3 - "$OUT" "$KSUB" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out, ksub = sys.argv[1], sys.argv[2]
for row in csv.DictReader(open(os.path.join(out, "trace", "trace_kernel_stats.csv"))):
    if ksub in row["Name"]:
        print("%-40s calls %s avg %.4f ms min %.4f max %.4f" % (row["Name"][:40], row["Calls"], float(row["AverageNs"]) / 1e6, float(row["MinNs"]) / 1e6, float(row["MaxNs"]) / 1e6))
seen = set()
for row in csv.DictReader(open(os.path.join(out, "trace", "trace_kernel_trace.csv"))):
    if ksub in row["Kernel_Name"] and row["Kernel_Name"] not in seen:
        seen.add(row["Kernel_Name"])
        print("  grid %s wg %s LDS %s VGPR %s accum %s SGPR %s scratch %s" % (row.get("Grid_Size"), row.get("Workgroup_Size"), row.get("LDS_Block_Size"), row.get("VGPR_Count"), row.get("Accum_VGPR_Count"), row.get("SGPR_Count"), row.get("Scratch_Size")))
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, "pmc*", "pmc_counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        if ksub in row["Kernel_Name"]:
            acc[row["Kernel_Name"][:48]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c in sorted(cs):
        v = cs[c]
        print("   %-26s max %.5g  (n=%d, sum %.5g)" % (c, max(v), len(v), sum(v)))
PY
