#!/bin/bash
# Round 6, call 10: side-stream lead stage (build in librp_engine_b.so): bitwise test + A/B.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r06_call10
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/robopianist_amd/csrc
RP_ENGINE_LIB=$L/librp_engine_b.so timeout 600 python -m pytest tests/test_gpu_env.py -m gpu -q -x -k "beside_the_chains or scripted_actions or step_after_last" > $R/pytest.log 2>&1; tail -4 $R/pytest.log
Q="--no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --aux-rccl 0"
run() { name=$1; shift; timeout 400 env "$@" > $R/$name.json 2> $R/$name.err; python -c "
import json,sys
d=json.loads([l for l in open('$R/$name.json').read().splitlines() if l.startswith('{\"metric\"')][-1]); print('$name', round(d['value']), round(d['ms_per_step'],3), d['roofline'].get('kernel_avg_ms'))"; }
for rep in 1 2 3; do
run A_$rep python bench.py $Q --steps 316
run B_$rep RP_ENGINE_LIB=$L/librp_engine_b.so python bench.py $Q --steps 316
run B0_$rep RP_ENGINE_LIB=$L/librp_engine_b.so RP_SIDE_LEAD=0 python bench.py $Q --steps 316
done
