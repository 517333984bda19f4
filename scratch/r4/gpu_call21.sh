#!/bin/bash
# What the order pass costs the step: the pass launched twice per substep (RP_X_ORDER_TWICE=1) against once.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r04_call21
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
FLAGS="--no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --config 2 --steps 158 --warmup 10 --fingertips hull"
for rep in 1 2 3; do for x in 0 1; do
  RP_X_ORDER_TWICE=$x timeout 300 python bench.py $FLAGS > $R/x${x}_$rep.json 2> $R/x${x}_$rep.err
  python -c "
import json
d=json.loads(open('$R/x${x}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']
print('config 2, order pass twice=$x #$rep value', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'seq', round(r['step_sequence_avg_ms'],3), 'sched', r['schedule'])"
done; done
