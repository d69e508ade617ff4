#!/bin/bash
# tools/build_file_variant.sh <name> <csrc basename> <extra hipcc flags...>: libu3d_hip variant that differs only in ONE source file
# (the other objects are the in-tree ones) -> uni3detr_amd/_variants/<name>.so
set -e
cd "$(dirname "$0")/.."
N=$1; F=$2; shift; shift
mkdir -p uni3detr_amd/_variants /tmp/fvar_$N
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Iinclude "$@" -c uni3detr_amd/csrc/$F.hip -o /tmp/fvar_$N/$F.o
OBJS=$(ls uni3detr_amd/csrc/_obj/*.o | grep -v -E "/$F\.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/fvar_$N/$F.o -o uni3detr_amd/_variants/$N.so
echo uni3detr_amd/_variants/$N.so
