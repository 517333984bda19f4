#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r04_call10
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
FLAGS="--no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --config 2 --steps 158 --warmup 10 --fingertips hull"
for rep in 1 2; do for lib in librp_engine librp_engine_keywalk; do
  RP_ENGINE_LIB=$PWD/robopianist_amd/csrc/$lib.so timeout 300 python bench.py $FLAGS > $R/${lib}_$rep.json 2> $R/${lib}_$rep.err
  python -c "
import json
d=json.loads(open('$R/${lib}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$lib #$rep value', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'seq', round(r['step_sequence_avg_ms'],3))"
done; done
timeout 400 python scratch/r4/solver_stats.py > $R/solver_stats.txt 2>&1; tail -7 $R/solver_stats.txt
