#!/bin/bash
# Round 6, call 6: slices x hardware queues on the split-stage schedule (config 2, 4096 envs).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r06_call6
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
Q="--no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --aux-rccl 0 --steps 316"
run() { name=$1; shift; timeout 400 env "$@" python bench.py $Q > $R/$name.json 2> $R/$name.err; python -c "
import json,sys
d=json.loads([l for l in open('$R/$name.json').read().splitlines() if l.startswith('{\"metric\"')][-1]); print('$name', round(d['value']), round(d['ms_per_step'],3), d['roofline'].get('schedule'))"; }
run default_a X=1
run s2_split RP_STREAM_SLICES=2 RP_SPLIT_POS=1 RP_FUSED=0
run s3_split RP_STREAM_SLICES=3 RP_SPLIT_POS=1 RP_FUSED=0
run s4_split RP_STREAM_SLICES=4 RP_SPLIT_POS=1 RP_FUSED=0
run s4_split_q8 GPU_MAX_HW_QUEUES=8 RP_STREAM_SLICES=4 RP_SPLIT_POS=1 RP_FUSED=0
run s3_split_q8 GPU_MAX_HW_QUEUES=8 RP_STREAM_SLICES=3 RP_SPLIT_POS=1 RP_FUSED=0
run s1_split RP_STREAM_SLICES=1 RP_SPLIT_POS=1 RP_FUSED=0
run s2_whole RP_STREAM_SLICES=2 RP_SPLIT_POS=0 RP_FUSED=0
run default_b X=1
