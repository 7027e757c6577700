# Kernel experiments on the GPU box: the parity tests that cover the rasterisers first, then the kernel times of
# alternative builds of the library side by side (LIBS: paths under renderer_amd/, e.g. "variants/libbase.so librdr_hip.so";
# a variant is the same objects linked with one file rebuilt differently, see profiles/sweep_lib.sh).
#   LIBS="variants/libbase.so librdr_hip.so" bash profiles/variant_check.sh
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 400 python3 -u -m pytest tests/test_gpu_parity.py tests/test_ref_raster.py tests/test_gpu_tile_path.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/exp_tests.log 2>&1 || { tail -30 gpurun_out/exp_tests.log; exit 1; }
tail -3 gpurun_out/exp_tests.log
for scene in field fieldxl atrium alphafield; do
  bash profiles/sweep_lib.sh "$LIBS" $scene 24
done
