set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
V=$GRAFT_REPO_ROOT/renderer_amd/variants
bash profiles/kstats.sh main_field field > $O/r3_ks6a.txt 2>&1 || exit 1
bash profiles/kstats.sh tri2_field field RDR_LIB_PATH=$V/librdr_tri2.so > $O/r3_ks6b.txt 2>&1 || exit 1
bash profiles/kstats.sh main_xl fieldxl > $O/r3_ks6c.txt 2>&1 || exit 1
bash profiles/kstats.sh tri2_xl fieldxl RDR_LIB_PATH=$V/librdr_tri2.so > $O/r3_ks6d.txt 2>&1 || exit 1
bash profiles/kstats.sh main_atrium atrium > $O/r3_ks6e.txt 2>&1 || exit 1
bash profiles/kstats.sh chunk64_atrium atrium RDR_LIB_PATH=$V/librdr_chunk64.so > $O/r3_ks6f.txt 2>&1 || exit 1
bash profiles/kstats.sh chunk8_atrium atrium RDR_LIB_PATH=$V/librdr_chunk8.so > $O/r3_ks6g.txt 2>&1 || exit 1
for f in main_field tri2_field main_xl tri2_xl; do echo "== $f"; grep "k_raster<false" $O/ks_$f.csv | awk -F, '{print $(NF-6), $(NF-5), $(NF-4)}'; done
for f in main_atrium chunk64_atrium chunk8_atrium; do echo "== $f"; grep "k_tile_setup\|k_mid_fill\|k_tile_raster" $O/ks_$f.csv | awk -F, '{print $1, $(NF-5), $(NF-4)}'; done
