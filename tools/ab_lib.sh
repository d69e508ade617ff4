#!/bin/bash
# same-box A/B of a library variant (uni3detr_amd/_variants/<name>.so, tools/build_file_variant.sh) against the in-tree build
cd "$GRAFT_REPO_ROOT"
V=$PWD/uni3detr_amd/_variants/$1.so
for i in 1 2; do
  echo "== base";    python tools/conv_bench.py --only dense256 --iters 20 --rotate 3 --data bn 2>&1 | grep -E "fwd  |dgrad"
  echo "== $1";      U3D_LIB_PATH=$V python tools/conv_bench.py --only dense256 --iters 20 --rotate 3 --data bn 2>&1 | grep -E "fwd  |dgrad"
done
echo "== parity of the variant (eight-phase tests run against it)"
U3D_LIB_PATH=$V python -m pytest tests/test_sparse_gpu.py -m gpu -q -k "eight_phase or dense_lattice or full_size or second3d" 2>&1 | tail -2
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-modes --no-workloads --no-roofline --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base', round(d['value'],1), round(d['ms_per_step'],3))"
  U3D_LIB_PATH=$V python bench.py --no-cpu-baseline --no-modes --no-workloads --no-roofline --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1  ', round(d['value'],1), round(d['ms_per_step'],3))"
done
