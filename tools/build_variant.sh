#!/bin/bash
# Builds a kernel variant of the library for A/B runs: tools/build_variant.sh <name> "<extra hipcc flags>"
# -> raytracer_amd/variants/librtmi_<name>.so (select with RTMI_LIB=...)
set -e
name=$1; shift
here=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $here/raytracer_amd/variants
cd $here/raytracer_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function $@ -shared -o ../variants/librtmi_$name.so rtmi_trace.hip rtmi_radiosity.hip rtmi_formfactor.hip rtmi_comm.hip rtmi_wavefront.hip rtmi_shade.hip bvh_build.cpp scene_io.cpp -lpthread -ldl
