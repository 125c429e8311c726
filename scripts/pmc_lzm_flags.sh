export TMPDIR=/tmp
for F in 0x77 0x67; do
  OUT=$PWD/gpurun_out/pmc_fl_$F; rm -rf $OUT; mkdir -p $OUT
  PNA_FLAGS=$F rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT -o sq -- python3 scripts/one_batch.py 2048 > $OUT.log 2>&1
  python3 - $OUT $F <<'PY'
import csv, glob, os, sys
from collections import defaultdict
res = defaultdict(dict)
for p in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(p)):
        k = row["Kernel_Name"].split("(")[0]
        res[k][row["Counter_Name"]] = res[k].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
for k, v in res.items():
    if "k_lzm" not in k: continue
    w = v.get("SQ_WAVES", 1) or 1
    print(sys.argv[2], k[:30], {n: round(x / w / 256, 1) for n, x in v.items() if n != "SQ_WAVES"})
PY
done
