#!/bin/bash
# same-box A/B of library variants on the captured bf16 step: tools/ab_step.sh <variant> [<variant> ...]  (uni3detr_amd/_variants/<variant>.so)
cd "$GRAFT_REPO_ROOT"
one() { python bench.py --no-cpu-baseline --no-modes --no-workloads --no-roofline --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],3))"; }
for i in 1 2 3; do
  one base
  for v in "$@"; do U3D_LIB_PATH=$PWD/uni3detr_amd/_variants/$v.so one $v; done
done
