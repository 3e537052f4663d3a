#!/bin/bash
# A/B of a compile-time switch of the kernels, with the numbers kept:
#   tools/ab.sh build <MACRO> <VALUE>     here (no GPU): raytracer_amd/variants/librtmi_<MACRO>_<VALUE>.so  (-D<MACRO>=<VALUE>)
#   tools/ab.sh run <name> [<name> ...]   on the GPU box (through gpurun): quick_trace / quick_c4 / quick_frame per build
#                                         ("base" = raytracer_amd/librtmi.so), logs under gpurun_out/ab/
#   tools/ab.sh rows                      here: one row per log appended to profiles/ab_r03.md
set -u
here=$(cd "$(dirname "$0")/.." && pwd)
case "${1:-}" in
build)
  bash $here/tools/build_variant.sh "$2_$3" "-D$2=$3" ;;
run)
  shift; mkdir -p $here/gpurun_out/ab
  for name in "$@"; do
    lib=$here/raytracer_amd/variants/librtmi_$name.so; [ "$name" = base ] && lib=$here/raytracer_amd/librtmi.so
    log=$here/gpurun_out/ab/$name.log; : > $log
    for tool in "quick_trace.py 5" "quick_c4.py" "quick_frame.py"; do
      RTMI_LIB=$lib TAG=$name timeout -k 10 200 python3 $here/tools/$tool >> $log 2>&1 || { echo "$name: $tool failed"; tail -n 3 $log; exit 1; }
    done
    echo "$name done"
  done ;;
rows)
  python3 - "$here" <<'PY'
import glob, os, re, sys
here = sys.argv[1]
rows = []
for f in sorted(glob.glob(os.path.join(here, "gpurun_out", "ab", "*.log"))):
    t = open(f).read()
    def g(pat, n=1):
        m = re.search(pat, t)
        return m.group(n) if m else "?"
    rows.append("| `%s` | %s | %s | %s | %s | %s | %s |" % (
        os.path.basename(f)[:-4], g(r"\| ([0-9.]+) ms  [0-9.]+ Mrays/s  digest"), g(r"ms  ([0-9.]+) Mrays/s  digest"), g(r"digest (\w+)"),
        g(r"shuffled rays via buffer: [0-9.]+ ms  ([0-9.]+) Mrays/s"), g(r"same bytes: \w+; frame ([0-9.]+) ms"), g(r"depth 1, AO 10: ([0-9.]+) ms/frame")))
out = os.path.join(here, "profiles", "ab_r03.md")
with open(out, "a") as fh:
    fh.write("\n## Traversal switches (`tools/ab.sh`: C3 frame, the same rays shuffled, C4 fused frame, studio frame)\n\n")
    fh.write("| build | C3 ms | C3 Mrays/s | frame digest | shuffled Mrays/s | C4 frame ms | studio depth-1 ms |\n|---|---|---|---|---|---|---|\n")
    fh.write("\n".join(rows) + "\n")
print("\n".join(rows))
PY
  ;;
*) echo "usage: tools/ab.sh build <MACRO> <VALUE> | run <name>... | rows"; exit 2 ;;
esac
