#!/bin/bash
# Smoke test of the measuring tools: each quick_*.py once, under its own time limit; stops at the first time-out or crash.
# usage (through gpurun): bash tools/run_all_quick.sh
set -u
for t in quick_assets quick_c2 quick_c4 quick_cast_one quick_display quick_ff_assets quick_ff_flat quick_jacobi quick_launch_cost quick_shard_solve quick_streams quick_tiles quick_trace; do
  echo "=== $t"
  timeout -k 10 150 python tools/$t.py > gpurun_out/tool_$t.log 2>&1
  rc=$?
  echo "rc=$rc"; tail -2 gpurun_out/tool_$t.log | cut -c1-220
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ] || [ $rc -eq 139 ]; then echo "stopping"; exit 1; fi
done
