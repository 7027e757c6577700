# The issue-rate microbenchmark, plain and once under the counters the issue fractions are computed from
# (calibrates SQ_INSTS_VALU x cycles / GRBM_GUI_ACTIVE on kernels whose SIMDs are known to be busy all the time).
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 $R/profiles/valu_rate > $O/valu_rate.txt
cat $O/valu_rate.txt
rm -rf $O/valu_pmc
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/valu_pmc -- $R/profiles/valu_rate > $O/valu_pmc.log 2>&1 || { tail -5 $O/valu_pmc.log; exit 1; }
python3 - <<PY
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/valu_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = open("$O/valu_pmc_summary.txt", "w")
for (k, d), c in sorted(rows.items(), key=lambda kv: int(kv[0][1])):
    s = {n: sum(v) for n, v in c.items()}
    line = "%-40s dispatch %3s  " % (k[:40], d) + "  ".join("%s %.0f" % (n, s[n]) for n in sorted(s))
    if s.get("GRBM_GUI_ACTIVE"):
        line += "  | INSTS_VALU*4/(1024*GRBM/8) = %.3f" % (s["SQ_INSTS_VALU"] * 4 / (1024 * s["GRBM_GUI_ACTIVE"] / 8))
    print(line, file=out)
out.close()
PY
tail -40 $O/valu_pmc_summary.txt
