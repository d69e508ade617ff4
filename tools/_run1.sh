cd /root/repo
for v in ms8 main ms2 ms8 main ms2; do
  if [ $v = main ]; then timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c60-200;
  else U3D_LIB_PATH=uni3detr_amd/_variants/$v.so timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c60-200; fi
done
