#!/bin/bash
# -fapprox-func (fast fp64 division: v_rcp + Newton steps instead of the IEEE sequence) against the shipped build.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r04_call23
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
FLAGS="--no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --config 2 --steps 158 --warmup 10"
for rep in 1 2; do for ft in hull primitive; do for lib in librp_engine librp_engine_afn; do
  RP_ENGINE_LIB=$PWD/robopianist_amd/csrc/$lib.so timeout 300 python bench.py $FLAGS --fingertips $ft > $R/${lib}_${ft}_$rep.json 2> $R/${lib}_${ft}_$rep.err
  python -c "
import json
d=json.loads(open('$R/${lib}_${ft}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$lib $ft #$rep value', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'seq', round(r['step_sequence_avg_ms'],3), 'sol', round(r['kernel_avg_ms'],4))"
done; done; done
RP_ENGINE_LIB=$PWD/robopianist_amd/csrc/librp_engine_afn.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_env.py -m gpu -q > $R/pytest_afn.log 2>&1; tail -4 $R/pytest_afn.log
