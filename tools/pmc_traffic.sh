#!/bin/bash
# HBM traffic of the dominant kernel from PMC counters (separate passes, as MI355X_MICROARCH.md prescribes).
# usage: tools/pmc_traffic.sh <tag>   -> gpurun_out/<tag>_{FETCH_SIZE,WRITE_SIZE}_igemm.csv
set -u
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for C in FETCH_SIZE WRITE_SIZE; do
  D=/tmp/pmc_${TAG}_$C
  rm -rf "$D"
  U3D_WATCHDOG_S=400 timeout 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$D" -o r -- python -W ignore bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph > gpurun_out/${TAG}_$C.log 2>&1
  f=$(find "$D" -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then head -1 "$f" > gpurun_out/${TAG}_${C}_igemm.csv; grep -E "k_igemm_glds|k_igemm_wgrad_glds|k_igemm_fwd|k_igemm_wgrad<" "$f" >> gpurun_out/${TAG}_${C}_igemm.csv; else echo "no counter csv" >> gpurun_out/${TAG}_$C.log; find "$D" | head >> gpurun_out/${TAG}_$C.log; fi
done
ls -la gpurun_out | grep "$TAG"
