#!/bin/bash
# Same-box A/B of the end-of-launch ray splitting (RTMI_SPLIT): C3 frame + shuffled rays, a rank's share of a dealt frame, C4 and studio frames.
# usage (on the GPU box): tools/ab_split.sh <log> [<variant name> ...]   ("base" = raytracer_amd/librtmi.so)
here=$(cd "$(dirname "$0")/.." && pwd)
log=$1; shift; : > $log
for name in "$@"; do
  lib=$here/raytracer_amd/variants/librtmi_$name.so; [ "$name" = base ] && lib=$here/raytracer_amd/librtmi.so
  echo "=== $name" >> $log
  for tool in "quick_trace.py 20" "quick_small_grid.py short" "quick_c4.py" "quick_frame.py"; do
    RTMI_LIB=$lib TAG=$name timeout -k 10 200 python3 -u $here/tools/$tool >> $log 2>&1 || { echo "$name: $tool failed"; tail -n 5 $log; exit 1; }
  done
done
grep -v amdgpu.ids $log
