# Kernel times of a workload under alternative builds of the library (RDR_LIB_PATH). Usage: bash profiles/sweep_lib.sh "lib1.so lib2.so ..." [scene] [frames]
R=${GRAFT_REPO_ROOT:-/root/repo}
SCENE=${2:-field}; FRAMES=${3:-24}
for l in $1; do
  echo "== $l $SCENE"
  RDR_LIB_PATH=$R/renderer_amd/$l timeout -k 10 150 python3 $R/profiles/run_frames.py $FRAMES $SCENE | grep -o "'raster': [0-9.]*\|'shadow': [0-9.]*\|'shade': [0-9.]*\|'big_raster': [0-9.]*" | tr '\n' ' ' || exit 1
  echo
done
