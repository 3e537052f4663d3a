#!/bin/bash
# Runs on the GPU box: texture-addresser / L1 (TA, TCP, TD) counters of the traversal kernels on the C3 frame (tools/quick_trace.py).
# usage: tools/prof_ta.sh <tag>   -> gpurun_out/prof_<tag>/ta*.  One rocprofv3 run per counter set, the program itself after "--".
set -u
OUT=gpurun_out/prof_$1
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for C in "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE" \
         "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TD_TD_BUSY_sum TD_TC_STALL_sum" \
         "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TD_LOAD_WAVEFRONT_sum"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $C --output-format csv -d $OUT/ta$i -o pmc -- python3 tools/quick_trace.py 3 > $OUT/ta$i.log 2>&1 || { echo "pass $i failed"; tail -n 3 $OUT/ta$i.log; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(sys.argv[1], "ta*", "pmc_counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        if "primary_closest" in row["Kernel_Name"] or "trace_rays" in row["Kernel_Name"]:
            acc[row["Kernel_Name"][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c in sorted(cs):
        v = cs[c]
        print("   %-40s mean %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
