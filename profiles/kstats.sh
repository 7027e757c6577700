# Per-kernel times of a few frames of one workload: bash profiles/kstats.sh <tag> <workload> [ENV=VALUE ...]
# Writes gpurun_out/ks_<tag>.csv (the rocprofv3 --kernel-trace --stats summary).
set -e
R=$GRAFT_REPO_ROOT
tag=$1; wl=$2; shift 2
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ks_$tag
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$tag -o $tag -- python3 $R/profiles/run_frames.py 30 $wl > $R/gpurun_out/ks_$tag.log 2>&1
f=$(find $R/gpurun_out/ks_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/ks_$tag.csv
cut -d, -f1-4 $R/gpurun_out/ks_$tag.csv | head -24
