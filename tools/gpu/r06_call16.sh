#!/bin/bash
# Round 6: eager stepping vs the captured hipGraph of one env.step on the final build (config 2, 4096 envs).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r06_call16; rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
Q="--no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --aux-rccl 0 --steps 316"
for rep in 1 2; do
for g in 0 1; do
timeout 400 python bench.py $Q --graph $g > $R/g${g}_$rep.json 2> $R/g${g}_$rep.err
python -c "
import json; d=json.loads(open('$R/g${g}_$rep.json').read().strip().splitlines()[-1]); print('graph $g', round(d['value']), round(d['ms_per_step'],3), d['roofline']['schedule'], d['config'].get('hipgraph_step'))"
done; done
