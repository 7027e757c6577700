set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > $O/r3_gputest4.log 2>&1; rc=$?; tail -6 $O/r3_gputest4.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 500 python -u profiles/rank_alone.py --frames 20 --out $O/r3_rank_alone2.json > $O/r3_rank_alone2.log 2>&1; tail -24 $O/r3_rank_alone2.log
