#!/bin/bash
# A/B inside one box on the random-policy configs (3, 5) and the headline: main library vs librp_engine_b.so.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r06_ab7
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
Q="--no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --aux-rccl 0"
run() { name=$1; shift; timeout 400 env "$@" > $R/$name.json 2> $R/$name.err; python -c "
import json,sys
d=json.loads([l for l in open('$R/$name.json').read().splitlines() if l.startswith('{\"metric\"')][-1]); print('$name', round(d['value']), round(d['ms_per_step'],3), d['roofline'].get('schedule'))" || tail -3 $R/$name.err; }
L=$GRAFT_REPO_ROOT/robopianist_amd/csrc
for rep in 1 2; do
for c in 3 5; do
run A_c${c}_$rep python bench.py $Q --config $c --steps 200
run B_c${c}_$rep RP_ENGINE_LIB=$L/librp_engine_b.so python bench.py $Q --config $c --steps 200
done
done
run A_c2 python bench.py $Q --steps 316
run B_c2 RP_ENGINE_LIB=$L/librp_engine_b.so python bench.py $Q --steps 316
