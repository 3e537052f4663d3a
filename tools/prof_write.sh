#!/bin/bash
# Runs on the GPU box: WRITE_SIZE / FETCH_SIZE of the traversal kernels on the C3 frame for the library RTMI_LIB names.  usage: tools/prof_write.sh <tag>
set -u
OUT=gpurun_out/prof_$1
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 120 python3 tools/quick_trace.py 10 2>/dev/null | grep Mrays || exit 1
i=0
for C in "WRITE_SIZE" "FETCH_SIZE"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $C --output-format csv -d $OUT/w$i -o pmc -- python3 tools/quick_trace.py 3 > $OUT/w$i.log 2>&1 || { echo "pass $i failed"; tail -n 3 $OUT/w$i.log; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(sys.argv[1], "w*", "pmc_counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        if "primary_closest" in row["Kernel_Name"]:
            acc[row["Kernel_Name"][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    for c in sorted(cs):
        v = cs[c]
        print("   %-20s mean %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
