#!/bin/bash
# Round 6, call 7: the fused schedule with the split stage's bodies (rp_fused_split_kernel) against the round-3 fused
# kernel and the per-stage schedules.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r06_call7
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
Q="--no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --aux-rccl 0"
run() { name=$1; shift; timeout 400 env "$@" > $R/$name.json 2> $R/$name.err; python -c "
import json,sys
d=json.loads([l for l in open('$R/$name.json').read().splitlines() if l.startswith('{\"metric\"')][-1]); print('$name', round(d['value']), round(d['ms_per_step'],3), d['roofline'].get('schedule'), d.get('sanity',{}).get('warn_flags_or'))"; }
run default python bench.py $Q --steps 316
run fusedsplit RP_FUSED=1 python bench.py $Q --steps 316
run fusedold RP_FUSED=1 RP_FUSED_SPLIT=0 python bench.py $Q --steps 316
run lock_default python bench.py $Q --steps 158 --stagger 0
run lock_fusedsplit RP_FUSED=1 python bench.py $Q --steps 158 --stagger 0
run lock_fusedold RP_FUSED=1 RP_FUSED_SPLIT=0 python bench.py $Q --steps 158 --stagger 0
run e2048_fusedsplit python bench.py $Q --envs 2048 --steps 158
run e2048_fusedold RP_FUSED_SPLIT=0 python bench.py $Q --envs 2048 --steps 158
run cap_fusedsplit RP_FUSED=1 python bench.py $Q --steps 316 --fingertips primitive
run cap_default python bench.py $Q --steps 316 --fingertips primitive
run e8192_fusedsplit RP_FUSED=1 python bench.py $Q --envs 8192 --steps 100
timeout 900 env RP_FUSED=1 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fused or split or replay_teacher or stream_slices or both_capacity" > $R/pytest_fused.log 2>&1; tail -3 $R/pytest_fused.log
