set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_bench $R/gpurun_out/prof_atrium $R/gpurun_out/pmc_field
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python3 $R/bench.py --steps 200 --warmup 20 --cpu-frames 0 --secondary-frames 0 > $R/gpurun_out/prof_bench.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_atrium -o atrium -- python3 $R/profiles/run_frames.py 20 atrium > $R/gpurun_out/prof_atrium.log 2>&1
for w in alphafield lodfield; do rm -rf $R/gpurun_out/prof_$w; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$w -o $w -- python3 $R/profiles/run_frames.py 20 $w > $R/gpurun_out/prof_$w.log 2>&1; done
bash $R/profiles/pmc_passes.sh $R/gpurun_out/pmc_field field
cd $R
python3 profiles/pmc_summary.py gpurun_out/pmc_field > gpurun_out/pmc_field4k_summary.txt
python3 profiles/pmc_traffic.py gpurun_out/pmc_field 1250 5 3840 2160 > gpurun_out/pmc_traffic.log
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
timeout -k 10 300 python3 profiles/parity_report.py > gpurun_out/parity_report.txt 2>&1
echo done
