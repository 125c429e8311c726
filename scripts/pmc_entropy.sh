set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_seq
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/a" -o s -- python3 bench.py --files 10000 --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-end-to-end > "$OUT/a.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d "$OUT/b" -o s -- python3 bench.py --files 10000 --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-end-to-end > "$OUT/b.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
res = defaultdict(dict)
for p in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(p)):
        k = row["Kernel_Name"].split("(")[0]
        res[k][row["Counter_Name"]] = res[k].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
for k, v in res.items():
    if "k_seq" in k or "k_lit" in k or "k_stats" in k or "k_write" in k or "k_frame" in k:
        w = v.get("SQ_WAVES", 1) or 1; wc = v.get("SQ_WAVE_CYCLES", 1) or 1
        print(k[:40], "waves", w, {n: round(x / w, 1) for n, x in v.items() if n.startswith("SQ_INSTS")}, {n: round(x / wc, 3) for n, x in v.items() if "ACTIVE" in n or "WAIT" in n}, "wave_cycles/wave", round(wc / w))
PY
