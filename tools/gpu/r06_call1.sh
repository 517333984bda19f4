#!/bin/bash
# Round 6, first GPU call: the GPU test suite, smoke, the default bench line (config 2, hull fingertips) and the per-phase
# shader-cycle profile of env 0 -- on the round-6 stand-in (impratio 10, wrist box clear of the palm).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r06_call1
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x > $R/pytest_gpu.log 2>&1; tail -15 $R/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $R/smoke.log 2>&1; tail -1 $R/smoke.log
timeout 1200 python bench.py > $R/bench_plain.json 2> $R/bench_plain.err; tail -c 1500 $R/bench_plain.json
RP_SCHED_DEBUG=1 timeout 600 python bench.py --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --steps 316 > $R/bench_short.json 2> $R/bench_short.err; grep "schedule choice" $R/bench_short.err | tail -3
timeout 600 python tools/gpu/phase_prof.py 64 4096 hull > $R/phase_prof.txt 2>&1; tail -60 $R/phase_prof.txt
