#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r04_call13
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
FLAGS="--no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --steps 158 --warmup 10 --fingertips hull"
for rep in 1 2; do for cfg in 2 3; do for sp in 0 1; do
  RP_SPLIT_HEAVY_POS=$sp timeout 300 python bench.py $FLAGS --config $cfg > $R/c${cfg}_s${sp}_$rep.json 2> $R/c${cfg}_s${sp}_$rep.err
  python -c "
import json
d=json.loads(open('$R/c${cfg}_s${sp}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']
print('config $cfg split $sp #$rep value', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'sol', round(r['kernel_avg_ms'],4), 'sched', r['schedule'])"
done; done; done
bash scratch/r4/ab_trees.sh r04_call13/ab_c2 --config 2 --steps 158 --warmup 10 --fingertips hull
