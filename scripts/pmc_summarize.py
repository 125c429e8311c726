#!/usr/bin/env python3
"""Sum FETCH_SIZE / WRITE_SIZE per kernel from rocprofv3 counter_collection CSVs (scripts/pmc_traffic.sh)."""
import csv, glob, json, os, sys
from collections import defaultdict

root, files = sys.argv[1], int(sys.argv[2])
res = defaultdict(lambda: {"launches": 0})
units = {}
for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    for p in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(p) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != ctr:
                    continue
                k = row["Kernel_Name"].split("(")[0]
                res[k][ctr] = res[k].get(ctr, 0.0) + float(row["Counter_Value"])
                if ctr == "FETCH_SIZE":
                    res[k]["launches"] += 1
print(json.dumps({"files": files, "note": "raw counter sums over all launches of one bench step; unit and gfx950 correction applied in profiles/README.md",
                  "kernels": res}, indent=1))
