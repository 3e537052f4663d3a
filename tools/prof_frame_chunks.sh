#!/bin/bash
# Per-kernel times of the studio frame (tools/prof_frame.py) and the C4 stages against the cap of the queue's claim size.
# usage (GPU box): tools/prof_frame_chunks.sh <outdir>
out=$1; mkdir -p $out; export TMPDIR=/tmp
for c in 256 512 1024 2048; do
  RTMI_MAX_CHUNK=$c timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/c$c -o t -- python3 tools/prof_frame.py 64 1 10 8 > $out/c$c.log 2>&1 || { echo "failed at $c"; tail -3 $out/c$c.log; exit 1; }
  echo "== RTMI_MAX_CHUNK=$c"
  python3 - "$out/c$c" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/t_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "rtmi" in r["Name"] and float(r["TotalDurationNs"]) > 2e5:
        print("  %-40s calls %3s avg %8.1f us min %8.1f" % (r["Name"].split("(")[0][-40:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
