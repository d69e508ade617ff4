cd /root/repo
timeout 900 python -m pytest tests/test_sparse_gpu.py -x -q -m gpu -k "bn_fwd_bwd" 2>&1 | tail -12
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_h.json 2> gpurun_out/bench_h.err; tail -c 400 gpurun_out/bench_h.err; cut -c1-420 gpurun_out/bench_h.json
