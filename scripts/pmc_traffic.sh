#!/bin/bash
# HBM traffic of every kernel from the L2 memory-side counters, one counter per pass (FETCH_SIZE costs 3 of the 4 TCC
# slots, WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC slots").  Run on the GPU box from the repo root:
#   bash scripts/pmc_traffic.sh [files]      -> gpurun_out/pmc/{fetch,write}/...counter_collection.csv
set -u
FILES=${1:-10000}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc
mkdir -p "$OUT"
for c in FETCH_SIZE WRITE_SIZE; do
  d=$OUT/$(echo $c | tr A-Z a-z | cut -d_ -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$d" -o pmc -- python3 bench.py --files "$FILES" --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-verify > "$d.log" 2>&1
  tail -1 "$d.log" | cut -c1-200
done
python3 scripts/pmc_summarize.py "$OUT" "$FILES" > "$OUT/summary.json"
cat "$OUT/summary.json"
