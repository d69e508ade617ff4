#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...>: experimental build of libu3d_hip into gpurun_out-free path uni3detr_amd/_variants/<name>.so
set -e
cd "$(dirname "$0")/.."
N=$1; shift
mkdir -p uni3detr_amd/_variants /tmp/var_$N
for f in uni3detr_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Iinclude "$@" -c $f -o /tmp/var_$N/$b.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/var_$N/*.o -o uni3detr_amd/_variants/$N.so
echo uni3detr_amd/_variants/$N.so
