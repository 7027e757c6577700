# Time split of k_shadow with the timing-only cuts (RDR_SHADOW_CUT: 1 = set-up alone, 2 = no top level). Usage: bash profiles/shadow_split.sh [scenes...]
R=${GRAFT_REPO_ROOT:-/root/repo}
for scene in "${@:-field}"; do
  for cut in 0 1 2; do
    echo "== $scene cut=$cut"
    RDR_SHADOW_CUT=$cut timeout -k 10 200 python3 $R/profiles/run_frames.py 24 $scene | grep -o "'shadow': [0-9.]*\|'shade': [0-9.]*" | tr '\n' ' ' || exit 1
    echo
  done
done
