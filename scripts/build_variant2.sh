#!/bin/bash
# Build a variant of libpna_gpu.so with extra -D flags on ONE kernel file: scripts/build_variant2.sh NAME FILE.hip -DFOO ...
# -> portable-network-archive_amd/variants/libpna_gpu_NAME.so (git-ignored; use with PNA_GPU_LIB)
set -e
NAME=$1; FILE=$2; shift; shift
D=portable-network-archive_amd/csrc
mkdir -p portable-network-archive_amd/variants
make -s -j8 -C $D
BASE=$(basename $FILE .hip)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c $D/$FILE -o /tmp/${BASE}_$NAME.o
OBJS=$(ls $D/*.o | grep -v "/$BASE.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o portable-network-archive_amd/variants/libpna_gpu_$NAME.so /tmp/${BASE}_$NAME.o $OBJS -Wl,-rpath,/opt/rocm/lib
echo built portable-network-archive_amd/variants/libpna_gpu_$NAME.so
