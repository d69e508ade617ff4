cd /root/repo
timeout 900 python -m pytest tests/test_sparse_gpu.py -x -q -m gpu -k "strided or lattice" 2>&1 | tail -5
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_k.json 2> gpurun_out/bench_k.err; tail -c 200 gpurun_out/bench_k.err; cut -c1-330 gpurun_out/bench_k.json
U3D_STRIDED_DGRAD_SPLIT=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-330
