#!/bin/bash
# kernel trace + stats only (primitive fingertips, no aux legs except the lockstep one), summarised on the box
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/prof_r02d
rm -rf $R; mkdir -p $R
cd /tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats -- $BENCH --steps 158 --warmup 20 > $R/stats.log 2>&1
cd $GRAFT_REPO_ROOT
python scratch/summarize_profiles_r02b.py $R 2>&1 | head -60
rm -rf $R/stats
