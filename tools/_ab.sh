cd /root/repo
for v in "$@"; do
  echo "== $v"
  U3D_LIB_PATH=uni3detr_amd/_variants/$v.so timeout 300 python tools/conv_bench.py --check 2>&1 | grep "dgrad\|rror"
done
