set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
bash profiles/kstats.sh atr_r0_norep atrium RUN_WORLD=8 RUN_RANK=0 RUN_REPLICATE=0 > $O/r3_ks_atr_r0_norep.txt 2>&1 || exit 1
bash profiles/kstats.sh atr_r7_norep atrium RUN_WORLD=8 RUN_RANK=7 RUN_REPLICATE=0 > $O/r3_ks_atr_r7_norep.txt 2>&1 || exit 1
bash profiles/kstats.sh atr_r0_rep atrium RUN_WORLD=8 RUN_RANK=0 RUN_REPLICATE=2048 > $O/r3_ks_atr_r0_rep.txt 2>&1 || exit 1
bash profiles/kstats.sh atr_r7_rep atrium RUN_WORLD=8 RUN_RANK=7 RUN_REPLICATE=2048 > $O/r3_ks_atr_r7_rep.txt 2>&1 || exit 1
for f in atr_r0_norep atr_r7_norep atr_r0_rep atr_r7_rep; do echo "== $f"; cut -d, -f1,2,4 $O/ks_$f.csv | head -14; done
