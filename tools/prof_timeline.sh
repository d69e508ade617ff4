#!/bin/bash
# usage: tools/prof_timeline.sh <tag> <command...>  -> gpurun_out/<tag>_timeline.txt (wall-time attribution of a replayed step, tools/timeline.py)
#        and gpurun_out/<tag>_kernel_stats.csv (rocprofv3 --stats of the same run)
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
D=/tmp/proftl_$TAG; rm -rf "$D"
rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -o r -- "$@" > gpurun_out/${TAG}_tl.log 2>&1
f=$(find "$D" -name '*kernel_trace.csv' | head -1)
s=$(find "$D" -name '*kernel_stats.csv' | head -1)
[ -n "$s" ] && cp "$s" gpurun_out/${TAG}_kernel_stats.csv
python tools/timeline.py "$f" gpurun_out/${TAG}_timeline.txt
head -75 gpurun_out/${TAG}_timeline.txt
