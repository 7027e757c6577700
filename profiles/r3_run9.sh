set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_sweep_slice.py tests/test_golden.py tests/test_gpu_tile_path.py tests/test_gpu_status.py -m gpu -x -q --timeout 200 --timeout-method thread > $O/r3_gputest9.log 2>&1; rc=$?; tail -4 $O/r3_gputest9.log; [ $rc -eq 0 ] || exit 1
bash profiles/kstats.sh row_atrium atrium > $O/r3_ks9a.txt 2>&1 || exit 1
bash profiles/kstats.sh row_xl fieldxl > $O/r3_ks9b.txt 2>&1 || exit 1
bash profiles/kstats.sh row_atr_r0 atrium RUN_WORLD=8 RUN_RANK=0 > $O/r3_ks9c.txt 2>&1 || exit 1
for f in row_atrium row_xl row_atr_r0; do echo "== $f"; grep "k_tile\|k_mid" $O/ks_$f.csv | awk -F, '{print $1, $(NF-5), $(NF-4)}'; done
