#!/bin/bash
# Variant of libpna_gpu.so with extra -D flags on ONE source: scripts/build_variant2.sh NAME file.hip -DFOO ...
set -e
NAME=$1; SRC=$2; shift; shift
D=portable-network-archive_amd/csrc
mkdir -p portable-network-archive_amd/variants
make -s -j8 -C $D
B=$(basename $SRC .hip)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c $D/$SRC -o /tmp/${B}_$NAME.o
OBJS=$(ls $D/*.o | grep -v "/$B.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o portable-network-archive_amd/variants/libpna_gpu_$NAME.so /tmp/${B}_$NAME.o $OBJS -Wl,-rpath,/opt/rocm/lib
echo built portable-network-archive_amd/variants/libpna_gpu_$NAME.so
