#!/bin/bash
# Round 6, call 4: GPU suite on the rule-based scheduler + bounding-capsule cull; half-filled batches under each schedule.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r06_call4
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x > $R/pytest_gpu.log 2>&1; tail -5 $R/pytest_gpu.log
Q="--no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --aux-rccl 0"
run() { name=$1; shift; timeout 400 "$@" > $R/$name.json 2> $R/$name.err; python -c "
import json,sys
d=json.loads(open('$R/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']), round(d['ms_per_step'],3), d['roofline'].get('schedule'), d['config']['solve_stats_last_step'])"; }
run c2_a python bench.py $Q --steps 316
run c2_b python bench.py $Q --steps 316
run c2_2048 python bench.py $Q --envs 2048 --steps 158
run c2_2048_s2 env RP_STREAM_SLICES=2 RP_FUSED=0 python bench.py $Q --envs 2048 --steps 158
run c2_2048_s1 env RP_STREAM_SLICES=1 RP_FUSED=0 python bench.py $Q --envs 2048 --steps 158
run c2_2048_s3 env RP_STREAM_SLICES=3 RP_FUSED=0 python bench.py $Q --envs 2048 --steps 158
run c5 python bench.py $Q --config 5 --steps 150
run c3 python bench.py $Q --config 3 --steps 150
