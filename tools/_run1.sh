cd /root/repo
timeout 900 python -m pytest tests/test_query_gpu.py -x -q -m gpu -k "colsum or safe_linear or fast_linear" 2>&1 | tail -15
timeout 600 python -m pytest tests/test_trainer_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -15
timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -3
