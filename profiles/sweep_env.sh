# Kernel times of a workload under values of one tuning variable. Usage: bash profiles/sweep_env.sh VAR "v1 v2 ..." [scene] [frames]
R=${GRAFT_REPO_ROOT:-/root/repo}
VAR=$1; VALUES=$2; SCENE=${3:-field}; FRAMES=${4:-24}
for v in $VALUES; do
  echo "== $VAR=$v $SCENE"
  env $VAR=$v timeout -k 10 150 python3 $R/profiles/run_frames.py $FRAMES $SCENE | grep -o "kernel_ms.: {[^}]*}" || exit 1
done
