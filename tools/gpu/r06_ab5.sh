#!/bin/bash
# A/B inside one box: main library vs librp_engine_b.so under several settings of an environment switch ($AB_VAR, values $AB_VALUES).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r06_ab5
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
Q="--no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --aux-rccl 0 ${AB_EXTRA}"
run() { name=$1; shift; timeout 400 env "$@" > $R/$name.json 2> $R/$name.err; python -c "
import json,sys
d=json.loads([l for l in open('$R/$name.json').read().splitlines() if l.startswith('{\"metric\"')][-1]); print('$name', round(d['value']), round(d['ms_per_step'],3), d['roofline'].get('kernel_avg_ms'), d['sanity']['warn_flags_or'], d['sanity']['finite'])" || tail -3 $R/$name.err; }
L=$GRAFT_REPO_ROOT/robopianist_amd/csrc
for rep in 1 2; do
run A_$rep python bench.py $Q --steps 316
for v in $AB_VALUES; do
run B_${v}_$rep RP_ENGINE_LIB=$L/librp_engine_b.so $AB_VAR=$v python bench.py $Q --steps 316
done
done
