#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r04_call20
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
timeout 900 python scratch/long_soak.py 20000 hull > $R/long_soak_hull.log 2>&1; tail -12 $R/long_soak_hull.log
