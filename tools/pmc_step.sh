#!/bin/bash
# Counter passes over one eager bench step of the CURRENT build, per (kernel, grid) averages:
#   tools/pmc_step.sh <tag>  -> gpurun_out/<tag>_{fetch,write,tcc,sq}_pmc.csv  (+ <tag>_trace_kernel_stats.csv: durations of the same command)
# Separate passes (FETCH_SIZE | WRITE_SIZE | TCC hit/miss | SQ), --kernel-trace only, as MI355X_MICROARCH.md prescribes.
set -u
TAG=$1
CMD="python -W ignore bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph"
RE='^(void )?k_'
tools/pmc_any.sh ${TAG}_fetch "FETCH_SIZE" "$RE" $CMD > /dev/null
tools/pmc_any.sh ${TAG}_write "WRITE_SIZE" "$RE" $CMD > /dev/null
tools/pmc_any.sh ${TAG}_tcc "TCC_HIT_sum TCC_MISS_sum" "$RE" $CMD > /dev/null
tools/pmc_any.sh ${TAG}_sq "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16" "$RE" $CMD > /dev/null
tools/prof_stats.sh ${TAG}_trace $CMD
ls -la gpurun_out | grep "$TAG"
