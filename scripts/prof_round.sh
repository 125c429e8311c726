#!/bin/bash
# Round-end evidence on the GPU box: kernel-trace stats of the default bench and of the encrypted variant, the bench lines next to them,
# and the HBM traffic of the cipher kernel (separate PMC passes).  Output under gpurun_out/prof_h/.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_h
mkdir -p "$OUT"
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python3 bench.py --encrypt aes-ctr --no-cpu-baseline > "$OUT/bench_aes_ctr.json" 2> "$OUT/bench_aes_ctr.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_default" -o kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/kt_default.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_aes" -o kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --encrypt aes-ctr > "$OUT/kt_aes.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  d=$OUT/pmc_$(echo $c | tr A-Z a-z | cut -d_ -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$d" -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify --encrypt aes-ctr > "$d.log" 2>&1
done
find "$OUT" -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head
