#!/bin/bash
# SQ counters for the igemm kernels: tools/pmc_sq.sh <tag> "<COUNTERS...>"
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
D=/tmp/pmcsq_$TAG; rm -rf "$D"
U3D_WATCHDOG_S=400 timeout 500 rocprofv3 --pmc $@ --kernel-trace --output-format csv -d "$D" -o r -- python -W ignore bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-graph > gpurun_out/${TAG}.log 2>&1
f=$(find "$D" -name '*counter_collection.csv' | head -1)
if [ -n "$f" ]; then head -1 "$f" > gpurun_out/${TAG}_igemm.csv; grep -E "k_igemm_glds_256x256|k_igemm_wgrad_glds_256|k_igemm_fwd<2, 4, 8, 4|k_igemm_wgrad<2, 4, 8, 4" "$f" >> gpurun_out/${TAG}_igemm.csv; else echo "no counter csv" >> gpurun_out/${TAG}.log; fi
