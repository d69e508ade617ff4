set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
python bench.py --steps 100 --warmup 10 > gpurun_out/r5a/bench_bf16.json 2> gpurun_out/r5a/bench_bf16.err
tail -c 600 gpurun_out/r5a/bench_bf16.err
python bench.py --steps 50 --warmup 5 --precision parity --no-roofline > gpurun_out/r5a/bench_parity.json 2> gpurun_out/r5a/bench_parity.err
tail -c 1500 gpurun_out/r5a/bench_parity.err
python bench.py --steps 50 --warmup 5 --precision fp32 --no-roofline --no-cpu-baseline > gpurun_out/r5a/bench_fp32.json 2> gpurun_out/r5a/bench_fp32.err
tail -c 1500 gpurun_out/r5a/bench_fp32.err
python -c "
import json
for n in ('bf16','parity','fp32'):
    try:
        d=json.loads(open('gpurun_out/r5a/bench_%s.json'%n).read().strip().splitlines()[-1])
        print(n, d['value'], d['ms_per_step'], d.get(n+'_vs_fp32_oracle'))
    except Exception as e: print(n, 'ERR', e)
"
