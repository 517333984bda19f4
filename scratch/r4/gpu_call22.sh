#!/bin/bash
# Sensitivity of the random-policy configs to the light class's entry capacity (RP_LEAN=n caps it at n entries; 184 = the build's).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r04_call22
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
FLAGS="--no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --steps 150 --warmup 10 --fingertips hull"
for cfg in 3 5; do for cap in 120 140 160 1; do
  RP_LEAN=$cap timeout 300 python bench.py $FLAGS --config $cfg > $R/c${cfg}_cap$cap.json 2> $R/c${cfg}_cap$cap.err
  python -c "
import json
d=json.loads(open('$R/c${cfg}_cap$cap.json').read().strip().splitlines()[-1]); r=d['roofline']
print('config $cfg light-class entry cap $cap (1 = 184) value', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'sol', round(r['kernel_avg_ms'],4))"
done; done
