#!/bin/bash
# Build a variant of libpna_gpu.so with extra -D flags on k_lz.hip and k_lz_split.hip only: scripts/build_variant.sh NAME -DFOO ...
# -> portable-network-archive_amd/variants/libpna_gpu_NAME.so (git-ignored; travels to the GPU box; use with PNA_GPU_LIB / scripts/ab.py)
set -e
NAME=$1; shift
D=portable-network-archive_amd/csrc
mkdir -p portable-network-archive_amd/variants
make -s -j8 -C $D
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c $D/k_lz.hip -o /tmp/k_lz_$NAME.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c $D/k_lz_split.hip -o /tmp/k_lz_split_$NAME.o
OBJS=$(ls $D/*.o | grep -v "k_lz.o\|k_lz_split.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o portable-network-archive_amd/variants/libpna_gpu_$NAME.so /tmp/k_lz_$NAME.o /tmp/k_lz_split_$NAME.o $OBJS -Wl,-rpath,/opt/rocm/lib
echo built portable-network-archive_amd/variants/libpna_gpu_$NAME.so
