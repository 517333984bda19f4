#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r04_call4
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
bash scratch/r4/ab_trees.sh r04_call4/ab_c2 --config 2 --steps 158 --warmup 10 --fingertips hull
bash scratch/r4/ab_trees.sh r04_call4/ab_c3 --config 3 --steps 150 --warmup 10 --fingertips hull
timeout 1500 python -m pytest tests -m gpu -q > $R/pytest_gpu.log 2>&1; tail -5 $R/pytest_gpu.log
