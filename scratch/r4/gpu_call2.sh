#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r04_call2
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q > $R/pytest_gpu.log 2>&1; tail -3 $R/pytest_gpu.log
bash scratch/r4/ab_trees.sh r04_call2/ab_c2 --config 2 --steps 158 --warmup 10 --fingertips hull
bash scratch/r4/ab_trees.sh r04_call2/ab_c3 --config 3 --steps 150 --warmup 10 --fingertips hull
