#!/bin/bash
# round 3, GPU call 3: GPU tests, kernel trace of the config 2 bench, per-phase profile
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_gpu.log 2>&1; tail -3 gpurun_out/r03/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof3 -o r03 -- python $GRAFT_REPO_ROOT/bench.py --config 2 --steps 60 --warmup 10 --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 > $GRAFT_REPO_ROOT/gpurun_out/r03/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof3 -name "*kernel_stats*" | head
f=$(find /tmp/prof3 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -12 "$f" | cut -c1-200 && cp "$f" gpurun_out/r03/kernel_stats.csv
tail -2 gpurun_out/r03/prof_bench.log | cut -c1-300
RP_SKIP_SELF_CHECK=1 timeout 300 python scratch/phase_prof.py 64 4096 > gpurun_out/r03/phase_prof.log 2>&1
tail -32 gpurun_out/r03/phase_prof.log
