#!/bin/bash
# Round 6, call 5: scripted replay inside the pre-step launch (test + bench), problem-size statistics on the new stand-in,
# config 2 at larger batches (slot-utilisation insight).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r06_call5
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_env.py -m gpu -q -x > $R/pytest_env.log 2>&1; tail -4 $R/pytest_env.log
Q="--no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --aux-rccl 0"
run() { name=$1; shift; timeout 400 "$@" > $R/$name.json 2> $R/$name.err; python -c "
import json,sys
d=json.loads([l for l in open('$R/$name.json').read().splitlines() if l.startswith('{\"metric\"')][-1]); print('$name', round(d['value']), round(d['ms_per_step'],3), d.get('step_sequence_avg_ms'), d['roofline'].get('schedule'))"; }
run c2_a python bench.py $Q --steps 316
run c2_b python bench.py $Q --steps 316
run c2_8192 python bench.py $Q --envs 8192 --steps 158
run c2_16384 python bench.py $Q --envs 16384 --steps 100
RP_STATS_OUT=$R/size_stats_hull.json timeout 600 python tools/gpu/size_stats.py hull > $R/size_stats_hull.log 2>&1; tail -3 $R/size_stats_hull.log
