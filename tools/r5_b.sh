set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
python -m pytest tests/test_postproc_gpu.py tests/test_bf16_parity_gpu.py tests/test_configs_gpu.py tests/test_trainer_gpu.py -x -q -m gpu 2>&1 | tail -8
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r5b/bench_default.json 2> gpurun_out/r5b/bench_default.err ) 2>&1 | tail -4
tail -c 400 gpurun_out/r5b/bench_default.err
python -c "
import json
d=json.loads(open('gpurun_out/r5b/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['cpu_baseline'])
for k,v in d['modes'].items(): print(k, {a:b for a,b in v.items() if a!='vs_fp32_oracle'})
"
