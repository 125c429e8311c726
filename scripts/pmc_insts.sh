#!/bin/bash
# Instruction mix of the LZ kernels per wave and 4 096-position tile (SQ_INSTS_*) of the product library: scripts/pmc_insts.sh [FILES]
# (k_lzm: 16 waves per segment, 256 tiles each; k_lzp: one wave per 128 KiB block = 32 tiles)
set -u
FILES=${1:-4096}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_insts
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM \
  --kernel-trace --output-format csv -d "$OUT" -o sq -- python3 bench.py --files "$FILES" --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-end-to-end > "$OUT.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
res = defaultdict(dict)
for p in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(p)):
        k = row["Kernel_Name"].split("(")[0]
        res[k][row["Counter_Name"]] = res[k].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
import json
out = {}
for k, v in res.items():
    if "k_lz" not in k: continue
    w = v.get("SQ_WAVES", 1) or 1
    tiles = 32 if "k_lzp" in k else 256
    for short in ("k_lzm", "k_lzp"):
        if short in k: out[short] = {n: round(x / w / tiles, 1) for n, x in v.items() if n != "SQ_WAVES"}
    json.dump(out, open(os.path.join(sys.argv[1], "insts.json"), "w"))
    print(k[:48], {n: round(x / w / tiles, 1) for n, x in v.items() if n != "SQ_WAVES"}, f"(per wave and 4096-position tile; {tiles} tiles per wave)", "waves", w)
PY
