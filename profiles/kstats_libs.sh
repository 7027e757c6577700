# rocprofv3 per-kernel averages of one workload under alternative builds of the library.
#   LIBS="variants/a.so librdr_hip.so" KERNELS="k_tile_setup|k_mid_fill" bash profiles/kstats_libs.sh atrium
set -e
R=$GRAFT_REPO_ROOT
wl=${1:-atrium}
cd /tmp && export TMPDIR=/tmp
for l in $LIBS; do
  tag=$(basename $l .so)
  rm -rf $R/gpurun_out/ksl_$tag
  RDR_LIB_PATH=$R/renderer_amd/$l timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ksl_$tag -o $tag -- python3 $R/profiles/run_frames.py 30 $wl > $R/gpurun_out/ksl_$tag.log 2>&1
  f=$(find $R/gpurun_out/ksl_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $l $wl"
  python3 - "$f" "${KERNELS:-k_}" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        n = r["Name"].replace("(anonymous namespace)::", "").replace("(SceneView, FrameView)", "").replace("void ", "")
        print("  %-36s %8.1f us" % (n[:36], float(r["AverageNs"]) / 1000))
PY
done
