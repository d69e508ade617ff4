cd /root/repo
timeout 900 python -m pytest tests/test_sparse_gpu.py -x -q -m gpu -k "epilogue_bn" 2>&1 | tail -8
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-330
