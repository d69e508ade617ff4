cd /root/repo
for v in "$@"; do
  echo "== $v"
  if [ "$v" = main ]; then timeout 300 python tools/conv_bench.py --only dense --check 2>&1 | grep "dense32\|dense16\|rror"; else
  U3D_LIB_PATH=uni3detr_amd/_variants/$v.so timeout 300 python tools/conv_bench.py --check 2>&1 | grep "dense32\|dense16\|rror"; fi
done
