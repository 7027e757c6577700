"""Summarise rocprofv3 --pmc CSV output per kernel: mean counter value per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::", "").replace("anonymous namespace)::", "")
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"    {c:36s} mean {sum(v) / len(v):16.1f}  n={len(v)}")
