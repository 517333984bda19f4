#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r04_call17
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
FLAGS="--no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --steps 158 --warmup 10 --fingertips hull"
for rep in 1 2; do for cfg in 2 3; do for so in 0 1; do
  RP_STALE_ORDER=$so timeout 300 python bench.py $FLAGS --config $cfg > $R/c${cfg}_o${so}_$rep.json 2> $R/c${cfg}_o${so}_$rep.err
  python -c "
import json
d=json.loads(open('$R/c${cfg}_o${so}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']
print('config $cfg stale order $so #$rep value', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'seq', round(r['step_sequence_avg_ms'],3), 'sol', round(r['kernel_avg_ms'],4), 'sched', r['schedule'])"
done; done; done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "slices or capacity_classes or fused or lazy or state_writes" > $R/pytest_sel.log 2>&1; tail -3 $R/pytest_sel.log
