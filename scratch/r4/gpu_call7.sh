#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r04_call7
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_env.py -m gpu -x -q > $R/pytest_env.log 2>&1; tail -5 $R/pytest_env.log
bash scratch/r4/ab_trees.sh r04_call7/ab_c2 --config 2 --steps 158 --warmup 10 --fingertips hull
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_env.py > $R/pytest_rest.log 2>&1; tail -5 $R/pytest_rest.log
