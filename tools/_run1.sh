cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 | cut -c1-700
