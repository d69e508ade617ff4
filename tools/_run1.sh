cd /root/repo
timeout 900 python -m pytest tests/test_trainer_gpu.py -x -q -m gpu 2>&1 | tail -5
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_j.json 2> gpurun_out/bench_j.err; tail -c 300 gpurun_out/bench_j.err; cut -c1-330 gpurun_out/bench_j.json
