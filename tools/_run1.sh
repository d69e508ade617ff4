cd /root/repo
U3D_FORCE_DDP=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-260
U3D_FORCE_DDP=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-graph 2>&1 | tail -1 | cut -c1-260
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3
