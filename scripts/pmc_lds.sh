#!/bin/bash
# The LDS unit under the LZ kernels: cycles it is busy with indexed operations, cycles stalled by bank / address conflicts, against the kernel's time.
#   bash scripts/pmc_lds.sh [files]  ->  gpurun_out/pmc_lds/summary.txt
set -u
FILES=${1:-4096}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_lds
rm -rf "$OUT"; mkdir -p "$OUT"
i=0
for grp in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ATOMIC_RETURN SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/p$i" -o lds -- python3 bench.py --files "$FILES" --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-verify > "$OUT/p$i.log" 2>&1
  tail -1 "$OUT/p$i.log" | cut -c1-100
done
python3 - "$OUT" "$FILES" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, os, sys
from collections import defaultdict
res = defaultdict(dict)
for p in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(p)):
        k = row["Kernel_Name"].split("(")[0]
        res[k][row["Counter_Name"]] = res[k].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
print(f"files {sys.argv[2]} x 1 MiB, one step; counters summed over all launches, XCDs and SEs; LDS cycles are per CU-cycle: / (GRBM_GUI_ACTIVE x 256 CUs)")
for k, v in sorted(res.items()):
    if "k_lz" not in k: continue
    gui = v.get("GRBM_GUI_ACTIVE", 0)
    print(k[:60], {n: f"{x:.4g}" for n, x in v.items()})
    if gui:
        cu = gui * 256 / 8          # GRBM_GUI_ACTIVE is summed over the 8 XCDs: per-XCD active cycles x 32 CUs each
        print("    LDS busy (IDX_ACTIVE) per CU-cycle %.3f, bank-conflict stall %.3f, address-conflict stall %.3f" %
              (v.get("SQ_LDS_IDX_ACTIVE", 0) / cu, v.get("SQ_LDS_BANK_CONFLICT", 0) / cu, v.get("SQ_LDS_ADDR_CONFLICT", 0) / cu))
PY
