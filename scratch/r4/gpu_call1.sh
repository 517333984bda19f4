#!/bin/bash
# Round 4, call 1: GPU tests + smoke + the default bench line on the build at the start of the session.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r04_call1
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q > $R/pytest_gpu.log 2>&1; tail -3 $R/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $R/smoke.log 2>&1; tail -1 $R/smoke.log
timeout 900 python bench.py > $R/bench_plain.json 2> $R/bench_plain.err; tail -c 600 $R/bench_plain.json
