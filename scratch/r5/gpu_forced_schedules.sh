#!/bin/bash
# The GPU parity + env suites under every schedule the engine may pick on its own, each forced from the environment.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r05_forced
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
for v in "RP_FUSED=1" "RP_FUSED=0 RP_STREAM_SLICES=2" "RP_LEAN=0" "RP_SPLIT_HEAVY_POS=0 RP_HEAVY_GRID=128" "RP_SPLIT_POS=1" "RP_SPLIT_POS=1 RP_STREAM_SLICES=3 RP_COMPANION=0 RP_FUSED=0" "RP_SPLIT_POS=0"; do
  n=$(echo $v | tr ' =' '__')
  env $v timeout 900 python -m pytest tests/test_gpu_env.py tests/test_gpu_parity.py -m gpu -q -x -k "not eight_ranks" > $R/pytest_$n.log 2>&1
  echo "$v: $(tail -1 $R/pytest_$n.log)"
done
