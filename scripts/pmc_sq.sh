#!/bin/bash
# SQ issue/wait breakdown of the kernels (one pass, 8 SQ slots).  Run on the GPU box from the repo root.
set -u
FILES=${1:-2048}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_sq
mkdir -p "$OUT"
rocprofv3 --pmc ${PMC_LIST:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES} \
  --kernel-trace --output-format csv -d "$OUT" -o sq -- python3 bench.py --files "$FILES" --steps 1 --warmup 0 --no-cpu-baseline > "$OUT.log" 2>&1
tail -1 "$OUT.log" | cut -c1-120
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
res = defaultdict(dict)
for p in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(p)):
        k = row["Kernel_Name"].split("(")[0]
        res[k][row["Counter_Name"]] = res[k].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
import json
issue = {}
for k, v in res.items():
    if "pna" not in k: continue
    wc = v.get("SQ_WAVE_CYCLES", 1) or 1
    print(k, {n: round(x / wc, 3) for n, x in v.items() if n != "SQ_WAVE_CYCLES"}, "wave_cycles", wc)
    # what bench.py's roofline.issue reports (profiles/r05_sq_issue.json): k_lzm runs 16 waves per CU = 4 per SIMD (160 KiB of LDS per workgroup), k_lz likewise
    for short in ("k_lzm", "k_lzp", "k_lz<"):
        if short in k and "ACTIVE" in "".join(v):
            issue[short.rstrip("<")] = {"kernel": k, "valu_active": round(v.get("SQ_ACTIVE_INST_VALU", 0) / wc, 4), "lds_active": round(v.get("SQ_ACTIVE_INST_LDS", 0) / wc, 4),
                                        "wait_any": round(v.get("SQ_WAIT_ANY", 0) / wc, 4), "wait_inst_any": round(v.get("SQ_WAIT_INST_ANY", 0) / wc, 4),
                                        "waves_per_simd": 4 if short != "k_lzp" else 4.5}
try:
    ins = json.load(open(os.path.join(os.path.dirname(sys.argv[1]), "pmc_insts", "insts.json")))
    for short, d in ins.items():
        if short in issue: issue[short]["valu_per_wave_tile"] = d.get("SQ_INSTS_VALU")
except Exception:
    pass
json.dump(issue, open(os.path.join(sys.argv[1], "issue.json"), "w"), indent=1)
PY
