#!/bin/bash
# usage (on the GPU box): tools/evidence.sh <tag>   -> the round's evidence set under gpurun_out/, to be copied into profiles/:
#   counter passes (tools/pmc_step.sh), the default bench line (with the modes block), the other workloads, the single-rank RCCL path,
#   the wall-time timeline and the per-grid launch durations of the implicit-GEMM kernels of a replayed step.
cd "$GRAFT_REPO_ROOT"
T=${1:-r05}
tools/pmc_step.sh $T > gpurun_out/${T}_pmc_step.log 2>&1; echo "pmc done"
timeout 900 python bench.py > gpurun_out/${T}_final_bench_line.json 2> gpurun_out/${T}_final_bench.err; echo "bench rc $?"
for c in kitti_3classes nuscenes scannet_large; do timeout 900 python bench.py --config $c --steps 40 --warmup 5 --no-cpu-baseline --no-workloads > gpurun_out/${T}_bench_line_$c.json 2> gpurun_out/${T}_bench_$c.err; echo "$c rc $?"; done
U3D_FORCE_DDP=1 timeout 900 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/${T}_bench_line_single_rank_rccl.json 2> gpurun_out/${T}_bench_rccl.err; echo "rccl rc $?"
tools/prof_timeline.sh ${T}_final python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --no-workloads --no-modes > /dev/null
tools/prof_trace_summary.sh ${T}_final k_igemm python bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline --no-workloads --no-modes > /dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/${T}_bench_line_*.json'))+['gpurun_out/${T}_final_bench_line.json']:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
        print(f.split('/')[-1], round(d['value'],1), round(d['ms_per_step'],2), 'frac', r.get('frac'), 'traffic', r.get('traffic'), (r.get('pmc_pass') or {}).get('stale_sources'), d['config']['launch_mode'][:24])
    except Exception as e: print(f,'ERR',e)
PY
