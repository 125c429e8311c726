#!/bin/bash
# Instruction mix (per wave) and issue / wait fractions of EVERY kernel of one bench.py step: scripts/pmc_kernels.sh TAG [bench.py arguments]
# Two counter passes, each with --kernel-trace only.  Run on the GPU box from the repo root.
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_kernels_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM \
  --kernel-trace --output-format csv -d "$OUT/insts" -o sq -- python3 bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-end-to-end > "$OUT/insts.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES \
  --kernel-trace --output-format csv -d "$OUT/waits" -o sq -- python3 bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-end-to-end > "$OUT/waits.log" 2>&1
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, os, sys
from collections import defaultdict
def load(sub):
    res = defaultdict(dict); calls = defaultdict(set)
    for p in glob.glob(os.path.join(sys.argv[1], sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(p)):
            k = row["Kernel_Name"].split("(")[0]
            res[k][row["Counter_Name"]] = res[k].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            calls[k].add(row.get("Dispatch_Id", ""))
    return res, calls
ins, calls = load("insts")
for k, v in sorted(ins.items()):
    w = v.get("SQ_WAVES", 1) or 1
    print(f"{k[:56]:56s} calls {len(calls[k]):3d} waves {int(w):9d} per wave:", {n[9:]: round(x / w, 1) for n, x in v.items() if n != "SQ_WAVES"})
wt, _ = load("waits")
for k, v in sorted(wt.items()):
    wc = v.get("SQ_WAVE_CYCLES", 1) or 1
    print(f"{k[:56]:56s} of wave cycles:", {n[3:]: round(x / wc, 3) for n, x in v.items() if n != "SQ_WAVE_CYCLES"}, "wave_cycles", int(wc))
PY
