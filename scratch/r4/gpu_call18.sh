#!/bin/bash
# The driver's own invocation (BENCH_r03.json: cmd), timed.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r04_call18
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  t0=$(date +%s.%N)
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $R/driver_like_$rep.json 2> $R/driver_like_$rep.err
  t1=$(date +%s.%N); echo "wall $(echo "$t1 - $t0" | bc) s"
  python -c "
import json
d=json.loads(open('$R/driver_like_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']
print('driver-like #$rep value', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'sched', r['schedule'], 'aux', {k: round(v['value']) for k, v in d['aux'].items() if isinstance(v, dict) and 'value' in v}, 'cpu', round(d['cpu_baseline']['value']))"
done
