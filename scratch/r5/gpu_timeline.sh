#!/bin/bash
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
R=$ROOT/gpurun_out/r05_timeline
rm -rf $R; mkdir -p $R
cd /tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --steps 40 --warmup 10"
env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/st -- $BENCH > $R/log 2>&1
python $ROOT/scratch/r5/timeline.py $R/st 90 | cut -c1-80
rm -rf $R/st
