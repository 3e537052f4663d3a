#!/bin/bash
# A/B of the form-factor kernel's dealing of samples (RTMI_FF_REGROUP*): box rows, the furnished room, the reference's meshes.
# usage (on the GPU box, through gpurun): tools/ab_regroup.sh <variant name | base> ...   -> gpurun_out/r3/regroup_<name>.log
here=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $here/gpurun_out/r3
for name in "$@"; do
  lib=$here/raytracer_amd/variants/librtmi_$name.so; [ "$name" = base ] && lib=$here/raytracer_amd/librtmi.so
  log=$here/gpurun_out/r3/regroup_$name.log; : > $log
  for tool in "quick_ff.py 4096" "quick_ff_mixed.py" "quick_ff_assets.py"; do
    RTMI_LIB=$lib timeout -k 10 300 python3 $here/tools/$tool >> $log 2>&1 || { echo "$name: $tool failed"; tail -n 5 $log; exit 1; }
  done
  echo "== $name"; grep -vE "amdgpu.ids" $log
done
