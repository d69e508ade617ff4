#!/bin/bash
# A/B of subm_halo.hip ablation builds (tools/build_file_variant.sh hl_<X> subm_halo -DHL_ABL_<X>): time only, results are wrong by design
cd "$(dirname "$0")/.."
for v in base "$@"; do
  if [ "$v" = base ]; then unset U3D_LIB_PATH; else export U3D_LIB_PATH=uni3detr_amd/_variants/$v.so; fi
  echo "== $v"; python tools/sparse_bench.py --only "64->64" --iters 50 2>&1 | grep -E "halo"
done
