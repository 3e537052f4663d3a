#!/bin/bash
# Re-sweep of the run-time knobs of trace_queue on the C3 frame (tools/quick_trace.py): refill threshold, vote weights.
# usage (GPU box): tools/sweep_queue.sh <log>
here=$(cd "$(dirname "$0")/.." && pwd)
log=$1; : > $log
run() { echo "=== $*" >> $log; env "$@" timeout -k 10 100 python3 -u $here/tools/quick_trace.py 20 2>&1 | grep -E "Mrays|shuffled" >> $log || exit 1; }
run RTMI_REFILL=16
run RTMI_REFILL=8
run RTMI_REFILL=12
run RTMI_REFILL=24
run RTMI_REFILL=32
run RTMI_VOTE_NODE=1 RTMI_VOTE_LEAF=2
run RTMI_VOTE_NODE=1 RTMI_VOTE_LEAF=4
run RTMI_VOTE_NODE=2 RTMI_VOTE_LEAF=5
run RTMI_MAX_CHUNK=256
cat $log
