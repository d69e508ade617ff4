cd /root/repo
timeout 900 python -m pytest tests/test_query_gpu.py -x -q -m gpu -k "layer_norm" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_trainer_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-330
timeout 800 bash tools/prof_stats.sh r01_m_graph_bf16 python bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline
grep "layernorm" gpurun_out/r01_m_graph_bf16_kernel_stats.csv | cut -c1-160
