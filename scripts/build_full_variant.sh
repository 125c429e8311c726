#!/bin/bash
# Build a variant of libpna_gpu.so with extra -D flags on EVERY source file: scripts/build_full_variant.sh NAME -DFOO ...
# -> portable-network-archive_amd/variants/libpna_gpu_NAME.so (git-ignored; use with PNA_GPU_LIB)
set -e
NAME=$1; shift
D=portable-network-archive_amd/csrc
O=/tmp/objs_$NAME
mkdir -p $O portable-network-archive_amd/variants
for f in $D/*.hip $D/*.cpp; do
  b=$(basename $f); b=${b%.*}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable "$@" -x hip -c $f -o $O/$b.o ) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o portable-network-archive_amd/variants/libpna_gpu_$NAME.so $O/*.o -Wl,-rpath,/opt/rocm/lib
echo built portable-network-archive_amd/variants/libpna_gpu_$NAME.so
