#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r04_call11
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
FLAGS="--no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --steps 150 --warmup 10 --fingertips hull"
for rep in 1 2; do for cfg in 3 4 2; do for pr in 0 1; do
  RP_HEAVY_PRIORITY=$pr timeout 300 python bench.py $FLAGS --config $cfg > $R/c${cfg}_p${pr}_$rep.json 2> $R/c${cfg}_p${pr}_$rep.err
  python -c "
import json
d=json.loads(open('$R/c${cfg}_p${pr}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']
print('config $cfg priority $pr #$rep value', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'sol', round(r['kernel_avg_ms'],4), 'overflow eps', d['sanity'].get('capacity_overflow_episodes'))"
done; done; done
