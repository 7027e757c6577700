"""HBM traffic per launch from the fetch / write PMC passes of pmc_passes.sh -> pmc_traffic.json.

  python profiles/pmc_traffic.py <pmc output dir> <instances> <subdivisions> <width> <height>

bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB: both counters tally KiB, and FETCH_SIZE is doubled
because gfx950 counts a 128-byte read request as 64 bytes.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
workload = [int(a) for a in sys.argv[2:6]]
acc = defaultdict(lambda: defaultdict(list))
for sub in ("fetch", "write"):
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
kernels = {}
for k, c in acc.items():
    if not k.startswith("k_") or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    fetch = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
    write = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
    kernels[k[2:]] = int((2.0 * fetch + write) * 1024)
# the bench's "raster" bracket holds both instantiations and the mid-size pass
merged = defaultdict(int)
for k, v in kernels.items():
    name = "raster" if k.startswith("raster<") or k == "mid_raster" else k.split("<")[0]  # template instantiations together
    merged[name] += v
out = {
    "workload": workload,
    "note": "HBM bytes per launch from rocprofv3 PMC passes (profiles/pmc_passes.sh fetch, write): "
            "(2*FETCH_SIZE + WRITE_SIZE) KiB; FETCH_SIZE doubled for gfx950's 64-B tally of 128-B reads; "
            "raster = k_raster<false> + k_raster<true> + k_mid_raster",
    "kernels": dict(sorted(merged.items())),
}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
