#!/bin/bash
# Round 6, call 2: what the full-capacity ("heavy") solver launch costs the 3-slice chain.  A/B inside one box:
#   default grid (8 ..) / RP_HEAVY_GRID=1,2 / the launch suppressed (RP_X_NO_HEAVY, experiments build: measurement only)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r06_call2
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --steps 316"
run() { name=$1; shift; env "$@" timeout 300 $B > $R/$name.json 2> $R/$name.err; python -c "
import json,sys
d=json.loads(open('$R/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']), d['ms_per_step'], d['roofline'].get('schedule'))"; }
for rep in 1 2; do
run default_$rep RP_SCHED_DEBUG=1
run grid1_$rep RP_HEAVY_GRID=1
run grid2_$rep RP_HEAVY_GRID=2
run noheavy_$rep RP_ENGINE_LIB=$GRAFT_REPO_ROOT/robopianist_amd/csrc/librp_engine_x.so RP_X_NO_HEAVY=1
run xlib_$rep RP_ENGINE_LIB=$GRAFT_REPO_ROOT/robopianist_amd/csrc/librp_engine_x.so
done
grep "schedule choice" $R/default_1.err | tail -2
# kernel trace of the schedule in use (timeline segment: the chain of one slice)
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/st -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --steps 100 --warmup 20 > $R/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/gpu/kstats.py $R/st 12
python tools/gpu/timeline.py $R/st 220 > $R/timeline.txt 2>&1; head -20 $R/timeline.txt
rm -rf $R/st
