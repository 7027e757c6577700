"""HBM traffic per launch from the fetch / write PMC passes of pmc_passes.sh -> pmc_traffic.json.

  python profiles/pmc_traffic.py <pmc output dir> <instances> <subdivisions> <width> <height>

bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB: both counters tally KiB, and FETCH_SIZE is doubled
because gfx950 counts a 128-byte read request as 64 bytes.

Also, from the sq3 and tcp passes, how close each kernel is to its instruction-issue bound:
simd_cycles_per_valu = (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) / SQ_INSTS_VALU, the SIMD cycles the launch had for
every wave64 vector instruction it issued, to be read against what an instruction costs (profiles/valu_rate.hip,
profiles/r3/valu_rate.txt: 2.3 - 2.8 cycles with register operands and four or more waves to pick from, 4.0 - 4.5
with a scalar-register operand, for 64-bit integer, packed fp32 and cross-lane instructions, 7.6 for a
transcendental; a wave alone on its SIMD gets one in every 5 - 9); valu_busy_lo / valu_busy_hi price every
instruction at 2.5 and at 4.2 cycles: the share of the launch's cycles in which its SIMDs were issuing vector
instructions lies between the two; and lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU).
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
    per_valu = (1024.0 * mean("GRBM_GUI_ACTIVE") / 8.0) / mean("SQ_INSTS_VALU")
    issue[k] = {"simd_cycles_per_valu": round(per_valu, 2),
                "valu_busy_lo": round(min(1.0, 2.5 / per_valu), 3), "valu_busy_hi": round(min(1.0, 4.2 / per_valu), 3),
                "lane_utilisation": round(mean("SQ_THREAD_CYCLES_VALU") / (64.0 * mean("SQ_INSTS_VALU")), 3)}
out = {
    "workload": workload,
    "note": "HBM bytes per launch from rocprofv3 PMC passes (profiles/pmc_passes.sh fetch, write): "
            "(2*FETCH_SIZE + WRITE_SIZE) KiB; FETCH_SIZE doubled for gfx950's 64-B tally of 128-B reads; "
            "raster = the k_raster instantiations + k_mid_fill",
    "kernels": dict(sorted(merged.items())),
    "issue_note": "simd_cycles_per_valu = (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs) / SQ_INSTS_VALU; a wave64 vector instruction costs a SIMD 2.3-2.8 cycles "
                  "with register operands, 4.0-4.5 with a scalar-register operand or as a 64-bit, packed or cross-lane instruction, 7.6 as a transcendental "
                  "(profiles/valu_rate.hip); valu_busy_lo / _hi price all of them at 2.5 / 4.2 cycles; "
                  "lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_INSTS_VALU) (passes sq3, tcp)",
    "issue": issue,
}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
