# Every test file that holds GPU tests, in a process of its own (order independence); stops at the first failure.
cd $GRAFT_REPO_ROOT
for f in $(grep -l "pytest.mark.gpu\|pytestmark = pytest.mark.gpu" tests/test_*.py | sort); do
  timeout -k 10 400 python3 -u -m pytest $f -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/each.log 2>&1 || { echo "FAILED $f"; tail -15 gpurun_out/each.log; exit 1; }
  echo "$f: $(tail -1 gpurun_out/each.log)"
done
