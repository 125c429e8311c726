#!/bin/bash
# Build variants of libpna_gpu.so whose LZ kernels carry timing experiments (PNA_EXP bits, k_lz_split.hip) into build/exp/ -- which travels to the GPU box --
# and, with `run`, time each on the box: scripts/exp_variants.sh build 1 2 4 8 ; gpurun -- 'bash scripts/exp_variants.sh run 0 1 2 4 8'
set -u
cd "$(dirname "$0")/.."
MODE=$1; shift
C=portable-network-archive_amd/csrc
mkdir -p build/exp
if [ "$MODE" = build ]; then
  for v in "$@"; do
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DPNA_EXP=$v -c $C/k_lz_split.hip -o build/exp/k_lz_split_$v.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp/libpna_$v.so $(ls $C/*.o | grep -v k_lz_split.o) build/exp/k_lz_split_$v.o -Wl,-rpath,/opt/rocm/lib -ldl && echo built $v ) &
  done
  wait
else
  FILES=${FILES:-4096}
  for v in "$@"; do
    lib=$PWD/build/exp/libpna_$v.so; [ "$v" = 0 ] && lib=$PWD/portable-network-archive_amd/libpna_gpu.so
    PNA_GPU_LIB=$lib python3 bench.py --files $FILES --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end ${EXTRA:-} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('variant $v', 'MiB/s', d['value'], 'ms/step', d['ms_per_step'], 'k_lzm', r['kernel_ms_per_step'], 'lz', r['lz_stage_ms'], 'ratio', d['ratio'], 'verified', d['verified'])"
  done
fi
