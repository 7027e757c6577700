set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > $O/r3_gputest3.log 2>&1; rc=$?; tail -6 $O/r3_gputest3.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --cpu-frames 0 > $O/r3_bench2.json 2> $O/r3_bench2.err || { tail -5 $O/r3_bench2.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_bench2.json'))
print('ms_per_step', d['ms_per_step'], d['kernel_ms'])
for s in d['secondary']: print(round(s['ms_per_frame'],4), s['workload'][:50])
PY
bash profiles/kstats.sh noatomics_field field RDR_LIB_PATH=$GRAFT_REPO_ROOT/renderer_amd/variants/librdr_atomics1.so > $O/r3_ks_noatomics_field.txt 2>&1 || exit 1
bash profiles/kstats.sh new_field field > $O/r3_ks_new_field.txt 2>&1 || exit 1
grep -h "k_raster\|k_tile\|k_mid" $O/r3_ks_noatomics_field.txt $O/r3_ks_new_field.txt
