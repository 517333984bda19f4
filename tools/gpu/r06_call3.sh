#!/bin/bash
# Round 6, call 3: the rule-based schedule choice on every BASELINE config (and the half-filled batch), graph replay.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r06_call3
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
Q="--no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --aux-rccl 0"
run() { name=$1; shift; timeout 400 "$@" > $R/$name.json 2> $R/$name.err; python -c "
import json,sys
d=json.loads(open('$R/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']), round(d['ms_per_step'],3), d['roofline'].get('schedule'), d.get('sanity'))"; }
run c2_a python bench.py $Q --steps 316
run c2_b python bench.py $Q --steps 316
run c2_graph python bench.py $Q --steps 316 --graph 1
run c2_lockstep python bench.py $Q --steps 158 --stagger 0
run c2_2048 python bench.py $Q --envs 2048 --steps 158
run c2_capsule python bench.py $Q --steps 316 --fingertips primitive
run c3 python bench.py $Q --config 3 --steps 150
run c4 python bench.py $Q --config 4 --steps 150
run c5 python bench.py $Q --config 5 --steps 150
# forced alternatives on config 3 / 4 (which rule would have been better)
run c3_s2 env RP_STREAM_SLICES=2 python bench.py $Q --config 3 --steps 150
run c3_s1 env RP_STREAM_SLICES=1 python bench.py $Q --config 3 --steps 150
run c4_s2 env RP_STREAM_SLICES=2 python bench.py $Q --config 4 --steps 150
run c4_s1 env RP_STREAM_SLICES=1 python bench.py $Q --config 4 --steps 150
run c5_s1 env RP_STREAM_SLICES=1 RP_FUSED=0 python bench.py $Q --config 5 --steps 150
run c5_s2 env RP_STREAM_SLICES=2 RP_FUSED=0 python bench.py $Q --config 5 --steps 150
run c2_rccl python bench.py --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --steps 60; python -c "import json;d=json.loads(open(\"$R/c2_rccl.json\").read().strip().splitlines()[-1]);print(d[\"aux\"][\"rccl_single_rank\"])"
timeout 300 python -m pytest tests/test_distributed.py -m gpu -q 2>&1 | tail -3
