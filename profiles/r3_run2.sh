set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
V=$GRAFT_REPO_ROOT/renderer_amd/variants/librdr_fastdiv.so
timeout -k 10 420 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > $O/r3_gputest2.log 2>&1; rc=$?; tail -4 $O/r3_gputest2.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --steps 200 --warmup 20 > $O/r3_bench1.json 2> $O/r3_bench1.err || { tail -5 $O/r3_bench1.err; exit 1; }
bash profiles/kstats.sh base_field field > $O/r3_ks_base_field.txt 2>&1 || exit 1
bash profiles/kstats.sh base_atrium atrium > $O/r3_ks_base_atrium.txt 2>&1 || exit 1
bash profiles/kstats.sh fast_field field RDR_LIB_PATH=$V > $O/r3_ks_fast_field.txt 2>&1 || exit 1
bash profiles/kstats.sh fast_atrium atrium RDR_LIB_PATH=$V > $O/r3_ks_fast_atrium.txt 2>&1 || exit 1
grep -h k_shade $O/r3_ks_*_*.txt
RDR_LIB_PATH=$V timeout -k 10 400 python -u -m pytest tests/test_gpu_sweep_slice.py tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_shading_kat.py tests/test_sampling_kat.py tests/test_golden.py tests/test_gltf.py tests/test_skybox.py -m gpu -q --timeout 200 --timeout-method thread > $O/r3_fastdiv_tests.log 2>&1; tail -15 $O/r3_fastdiv_tests.log
RDR_LIB_PATH=$V SWEEP_PART=all timeout -k 10 500 python -u profiles/parity_sweep.py > $O/r3_fastdiv_sweep.log 2>&1; grep -E "MISMATCH|done" $O/r3_fastdiv_sweep.log | head -20
