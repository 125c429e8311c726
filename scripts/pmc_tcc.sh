#!/bin/bash
# L2 (TCC) side of the LZ kernels: requests to memory by size, hits and misses -- what settles "bound by issue or by HBM" for k_lzp (one counter group per pass:
# the TCC has four slots).  Run on the GPU box from the repo root:   bash scripts/pmc_tcc.sh [files]  ->  gpurun_out/pmc_tcc/summary.txt
set -u
FILES=${1:-4096}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_tcc
rm -rf "$OUT"; mkdir -p "$OUT"
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/p$i" -o tcc -- python3 bench.py --files "$FILES" --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-verify > "$OUT/p$i.log" 2>&1
  tail -1 "$OUT/p$i.log" | cut -c1-100
done
python3 - "$OUT" "$FILES" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, os, sys
from collections import defaultdict
res = defaultdict(dict); dur = defaultdict(float)
for p in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(p)):
        k = row["Kernel_Name"].split("(")[0]
        res[k][row["Counter_Name"]] = res[k].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
for p in glob.glob(os.path.join(sys.argv[1], "p1", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(p)):
        dur[row["Kernel_Name"].split("(")[0]] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
print(f"files {sys.argv[2]} x 1 MiB, one step; counters summed over a kernel's launches; ms = kernel time in the counter pass (slower than an unprofiled run)")
for k, v in sorted(res.items()):
    if "k_lz" not in k: continue
    rd = v.get("TCC_EA0_RDREQ_sum", 0); rd32 = v.get("TCC_EA0_RDREQ_32B_sum", 0)
    rbytes = rd32 * 32 + (rd - rd32) * 64
    wr = v.get("TCC_EA0_WRREQ_sum", 0); wr64 = v.get("TCC_EA0_WRREQ_64B_sum", 0)
    wbytes = wr64 * 64 + (wr - wr64) * 32
    hit, miss = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
    ms = dur.get(k, 0)
    print(f"{k[:60]:60s} ms {ms:8.3f}  EA read {rbytes/1e9:8.2f} GB (req {rd:.3e}, 32B {rd32:.3e})  EA write {wbytes/1e9:8.2f} GB (req {wr:.3e}, 64B {wr64:.3e})  "
          f"L2 hit {hit:.3e} miss {miss:.3e} hit-rate {hit/max(hit+miss,1):.3f}  req {v.get('TCC_REQ_sum',0):.3e} read {v.get('TCC_READ_sum',0):.3e} write {v.get('TCC_WRITE_sum',0):.3e}"
          + (f"  -> memory side {(rbytes+wbytes)/1e9/(ms/1e3)/1e3:.2f} TB/s" if ms else ""))
PY
