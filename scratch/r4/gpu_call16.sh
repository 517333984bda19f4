#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r04_call16
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_bench_cli.py -m gpu -x -q -s -k "eight_ranks" > $R/pytest_8ranks.log 2>&1; grep "8 ranks\|passed\|failed" $R/pytest_8ranks.log | tail -3
bash scratch/r4/ab_trees.sh r04_call16/ab_c2 --config 2 --steps 158 --warmup 10 --fingertips hull
