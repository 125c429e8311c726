#!/bin/bash
# Instruction mix of the kernels per wave (SQ_INSTS_*), for one or more library builds: scripts/pmc_insts.sh FILES lib1.so lib2.so ...
set -u
FILES=${1:-1024}; shift
export TMPDIR=/tmp
for LIB in "$@"; do
  NAME=$(basename "$LIB" .so)
  OUT=$PWD/gpurun_out/pmc_insts_$NAME
  rm -rf "$OUT"; mkdir -p "$OUT"
  export PNA_GPU_LIB=$PWD/$LIB
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM \
    --kernel-trace --output-format csv -d "$OUT" -o sq -- python3 bench.py --files "$FILES" --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-end-to-end > "$OUT.log" 2>&1
  python3 - "$OUT" "$NAME" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
res = defaultdict(dict)
for p in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(p)):
        k = row["Kernel_Name"].split("(")[0]
        res[k][row["Counter_Name"]] = res[k].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
for k, v in res.items():
    if "k_lz" not in k: continue
    w = v.get("SQ_WAVES", 1) or 1
    print(sys.argv[2], k[:40], {n: round(x / w / 256, 1) for n, x in v.items() if n != "SQ_WAVES"}, "(per wave and 4096-position tile)", "waves", w)
PY
done
