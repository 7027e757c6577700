#!/bin/bash
# Counter passes (one rocprofv3 run each; never combined with tracing). Usage: pmc_passes.sh <outdir> [scene]
set -u
OUT=$1; SCENE=${2:-field}; ONLY=${3:-all}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT
run() { # name counters...
  name=$1; shift
  if [ "$ONLY" != "all" ] && [[ ",$ONLY," != *",$name,"* ]]; then return 0; fi
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/profiles/run_frames.py 3 $SCENE > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; return 1; }
  echo "pass $name ok"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU &&
run sq2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR &&
run tcp TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TA_TA_BUSY TA_FLAT_READ_WAVEFRONTS GRBM_GUI_ACTIVE &&
run fetch FETCH_SIZE TCC_REQ &&
run write WRITE_SIZE TCC_HIT TCC_MISS &&
run sq3 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
