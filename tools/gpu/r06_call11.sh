#!/bin/bash
# Round 6: per-phase shader-clock profile of env 0 (fp64, hull, three-slice schedule and one-slice one-kernel schedule) +
# the headline workload at 8192 / 16384 envs per GPU on the final build.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r06_call11; rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
timeout 300 python tools/gpu/phase_prof.py 64 4096 hull > $R/phase_prof_default.txt 2>&1
RP_STREAM_SLICES=1 RP_SPLIT_POS=0 timeout 300 python tools/gpu/phase_prof.py 64 4096 hull > $R/phase_prof_1slice.txt 2>&1
B="python bench.py --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --aux-rccl 0 --steps 158 --warmup 20"
for e in 8192 16384; do timeout 600 $B --envs $e > $R/bench_$e.json 2> $R/bench_$e.err; python -c "
import json,sys; d=json.loads(open('$R/bench_$e.json').read().strip().splitlines()[-1]); print($e, round(d['value']), d['ms_per_step'], d['roofline']['schedule'])"; done
cat $R/phase_prof_1slice.txt
