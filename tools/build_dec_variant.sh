#!/bin/bash
# tools/build_dec_variant.sh <name> <extra hipcc flags...>: libu3d_hip variant that differs only in decoder.hip / decoder_bwd.hip
# (the other objects are the in-tree ones) -> uni3detr_amd/_variants/<name>.so
set -e
cd "$(dirname "$0")/.."
N=$1; shift
mkdir -p uni3detr_amd/_variants /tmp/dvar_$N
for b in decoder decoder_bwd; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Iinclude "$@" -c uni3detr_amd/csrc/$b.hip -o /tmp/dvar_$N/$b.o &
done
wait
OBJS=$(ls uni3detr_amd/csrc/_obj/*.o | grep -v -E "/decoder(_bwd)?\.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/dvar_$N/decoder.o /tmp/dvar_$N/decoder_bwd.o -o uni3detr_amd/_variants/$N.so
echo uni3detr_amd/_variants/$N.so
