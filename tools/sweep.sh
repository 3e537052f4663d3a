#!/bin/bash
# usage: tools/sweep.sh "VAR=val VAR2=val" "..." ...   (each argument = one environment for tools/quick_trace.py)
for cfg in "$@"; do
  env $cfg TAG="$cfg" timeout -k 10 120 python3 tools/quick_trace.py 3 || exit 1
done
