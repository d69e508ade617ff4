#!/bin/bash
# usage: tools/prof_stats.sh <tag> <command...>   -> gpurun_out/<tag>_kernel_stats.csv (+ the command's log)
# rocprofv3 kernel trace + stats; only the per-kernel summary is kept (the full trace exceeds gpurun's 64 MiB return cap).
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
D=/tmp/prof_$TAG
rm -rf "$D"
rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -o r -- "$@" > gpurun_out/${TAG}.log 2>&1
f=$(find "$D" -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cp "$f" gpurun_out/${TAG}_kernel_stats.csv; else echo "no stats csv found" >> gpurun_out/${TAG}.log; find "$D" | head -20 >> gpurun_out/${TAG}.log; fi
