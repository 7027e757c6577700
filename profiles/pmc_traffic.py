"""HBM traffic per launch from the fetch / write PMC passes of pmc_passes.sh -> pmc_traffic.json.

  python profiles/pmc_traffic.py <pmc output dir> <instances> <subdivisions> <width> <height>

bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB: both counters tally KiB, and FETCH_SIZE is doubled
because gfx950 counts a 128-byte read request as 64 bytes.

Also, from the sq3 and tcp passes, how close each kernel is to its instruction-issue bound:
valu_busy = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): a SIMD of this chip takes 4 cycles
per wave64 vector instruction however many waves it has to pick from (profiles/valu_rate.hip: 4.4 cycles per
v_fma_f32 with one wave per SIMD, 3.9 - 4.1 with two to eight; a packed v_pk_fma_f32 takes the same 4 cycles for
twice the arithmetic), so this is the share of the launch's cycles in which its SIMDs were issuing vector
instructions -- a lower bound, transcendental and 64-bit instructions take longer; and
lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU).
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
    name = "raster" if k.startswith("raster<") or k == "mid_fill" else k.split("<")[0]  # template instantiations together
    merged[name] += v
issue_acc = defaultdict(lambda: defaultdict(list))
for sub in ("sq3", "tcp"):
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if k.startswith("k_"):
                # the opaque instantiation stands for the bench's "raster" bracket, the others keep their full names
                name = k[2:].split("<")[0] if not k.startswith("k_raster<") else ("raster" if k.startswith("k_raster<false") else k[2:])
                issue_acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
issue = {}
for k, c in sorted(issue_acc.items()):
    if not all(n in c for n in ("SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE")):
        continue
    mean = lambda n: sum(c[n]) / len(c[n])
    if mean("GRBM_GUI_ACTIVE") <= 0 or mean("SQ_INSTS_VALU") <= 0:
        continue
    issue[k] = {"valu_busy": round(mean("SQ_INSTS_VALU") * 4.0 / (1024.0 * mean("GRBM_GUI_ACTIVE") / 8.0), 3),
                "lane_utilisation": round(mean("SQ_THREAD_CYCLES_VALU") / (64.0 * mean("SQ_INSTS_VALU")), 3)}
out = {
    "workload": workload,
    "note": "HBM bytes per launch from rocprofv3 PMC passes (profiles/pmc_passes.sh fetch, write): "
            "(2*FETCH_SIZE + WRITE_SIZE) KiB; FETCH_SIZE doubled for gfx950's 64-B tally of 128-B reads; "
            "raster = the k_raster instantiations + k_mid_fill",
    "kernels": dict(sorted(merged.items())),
    "issue_note": "valu_busy = SQ_INSTS_VALU * 4 cycles (a SIMD issues one wave64 vector instruction per 4 cycles at any occupancy: profiles/valu_rate.hip) / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs); "
                  "lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_INSTS_VALU) (passes sq3, tcp)",
    "issue": issue,
}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
