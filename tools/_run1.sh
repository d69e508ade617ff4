cd /root/repo
for i in 1 2; do
U3D_FAST_LINEAR=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-200
U3D_FAST_LINEAR=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-200
done
