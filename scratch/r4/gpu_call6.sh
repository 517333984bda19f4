#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r04_call6
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
bash scratch/r4/ab_trees.sh r04_call6/ab_c2 --config 2 --steps 158 --warmup 10 --fingertips hull
bash scratch/r4/ab_trees.sh r04_call6/ab_c3 --config 3 --steps 150 --warmup 10 --fingertips hull
FLAGS="--no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --config 3 --steps 150 --warmup 10 --fingertips hull"
for g in 128 48 24; do
  RP_HEAVY_GRID=$g timeout 300 python bench.py $FLAGS > $R/c3_grid$g.json 2> $R/c3_grid$g.err
  python -c "
import json,sys
d=json.loads(open('$R/c3_grid$g.json').read().strip().splitlines()[-1]); print('config3 fixed heavy grid $g value', round(d['value']), 'ms/step', round(d['ms_per_step'],3))"
done
timeout 300 python scratch/phase_prof.py 64 4096 hull > $R/phase_new.txt 2>&1
grep "narrow geometry\|total cycles\|trace\|contact emission" $R/phase_new.txt
