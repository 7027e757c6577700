# Kernel times of a workload under settings of one environment variable.
#   VAR=RDR_TILE_FLAGS VALUES="0 1" SCENES="field atrium" bash profiles/env_sweep.sh
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
for scene in ${SCENES:-field}; do
  for v in $VALUES; do
    echo "== $VAR=$v $scene"
    env $VAR=$v timeout -k 10 150 python3 $R/profiles/run_frames.py 24 $scene | grep -o "'[a-z_]*': 0\.[0-9]*" | tr '\n' ' '
    echo
  done
done
