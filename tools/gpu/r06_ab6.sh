#!/bin/bash
# A/B inside one box: main library vs librp_engine_b.so on config 2 at 4096 and at 2048 envs (the fused-substeps schedule),
# with the per-stage schedule forced at 2048 as a third column.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r06_ab6
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
Q="--no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --aux-rccl 0"
run() { name=$1; shift; timeout 400 env "$@" > $R/$name.json 2> $R/$name.err; python -c "
import json,sys
d=json.loads([l for l in open('$R/$name.json').read().splitlines() if l.startswith('{\"metric\"')][-1]); print('$name', round(d['value']), round(d['ms_per_step'],3), d['roofline'].get('schedule'))" || tail -3 $R/$name.err; }
L=$GRAFT_REPO_ROOT/robopianist_amd/csrc
for rep in 1 2; do
run A4096_$rep python bench.py $Q --steps 316
run B4096_$rep RP_ENGINE_LIB=$L/librp_engine_b.so python bench.py $Q --steps 316
run A2048_$rep python bench.py $Q --steps 316 --envs 2048
run B2048_$rep RP_ENGINE_LIB=$L/librp_engine_b.so python bench.py $Q --steps 316 --envs 2048
run B2048_nofuse_$rep RP_ENGINE_LIB=$L/librp_engine_b.so RP_FUSED=0 python bench.py $Q --steps 316 --envs 2048
done
run B_fp32 RP_ENGINE_LIB=$L/librp_engine_b.so python bench.py $Q --steps 158 --precision 32
run A_fp32 python bench.py $Q --steps 158 --precision 32
