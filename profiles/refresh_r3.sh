# Round-3 measurement set, run on the GPU box from the repository root: bash profiles/refresh_r3.sh
# Writes under gpurun_out/r3prof/; the summaries are then copied into profiles/r3/ and committed.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# kernel times of the very command the bench line comes from
rm -rf $O/prof_bench
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python3 $R/bench.py --steps 200 --warmup 20 --cpu-frames 0 --secondary-frames 0 > $O/bench_profiled.json 2> $O/bench_profiled.err
cp $(find $O/prof_bench -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bench_field4k.csv
for w in atrium alphafield lodfield fieldxl; do
  rm -rf $O/prof_$w
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -o $w -- python3 $R/profiles/run_frames.py 30 $w > $O/prof_$w.log 2>&1
  cp $(find $O/prof_$w -name "*kernel_stats.csv" | head -1) $O/kernel_stats_${w}.csv
done
# counters: one pass per rocprofv3 run, never together with tracing
bash $R/profiles/pmc_passes.sh $O/pmc_field field
bash $R/profiles/pmc_passes.sh $O/pmc_atrium atrium sq3,tcp,fetch,write
cd $R
python3 profiles/pmc_summary.py $O/pmc_field > $O/pmc_field4k_summary.txt
python3 profiles/pmc_summary.py $O/pmc_atrium > $O/pmc_atrium4k_summary.txt
python3 profiles/pmc_traffic.py $O/pmc_field 1250 5 3840 2160 > $O/pmc_traffic.log
cp profiles/pmc_traffic.json $O/pmc_traffic.json
# the rasteriser's own ceiling, the split's balance, the default bench line, and a two-rank rehearsal on the one GPU
timeout -k 10 120 profiles/valu_rate > $O/valu_rate.txt
timeout -k 10 120 python3 profiles/atomic_peak.py > $O/atomic_peak.json
timeout -k 10 500 python3 -u profiles/rank_alone.py --frames 20 --out $O/rank_alone.json > $O/rank_alone.txt 2>&1
timeout -k 10 400 python3 bench.py > $O/bench_default_n1.json 2> $O/bench_default_n1.err
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --steps 100 --warmup 10 > $O/bench_gloo_2ranks_one_gpu.json 2> $O/bench_gloo_2ranks_one_gpu.err || echo "two-rank rehearsal failed"
echo refresh done
