cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_n.json 2> gpurun_out/bench_n.err; tail -c 200 gpurun_out/bench_n.err; cat gpurun_out/bench_n.json
