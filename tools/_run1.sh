cd /root/repo
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_i.json 2> gpurun_out/bench_i.err; cut -c1-330 gpurun_out/bench_i.json
timeout 800 bash tools/prof_stats.sh r01_i_graph_bf16 python bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline
