#!/bin/bash
# Register / scratch / LDS use of every kernel in one .hip source, read off the gfx950 code object's metadata.
# usage: tools/kernel_resources.sh raytracer_amd/csrc/rtmi_formfactor.hip [extra hipcc flags]   (no GPU needed)
set -e
src=$1; shift
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off --cuda-device-only "$@" -c "$src" -o $tmp/k.co
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$tmp/k.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/k.elf
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/k.elf | grep -E "^\s+\.name:|private_segment_fixed_size|\.vgpr_count|\.sgpr_count|group_segment_fixed|\.vgpr_spill_count|\.sgpr_spill_count" | paste - - - - - - - | sed -e 's/  */ /g'
if [ -n "$KEEP" ]; then /opt/rocm/lib/llvm/bin/llvm-objdump -d $tmp/k.elf > "$KEEP"; fi
rm -rf $tmp
