#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r04_call19
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -rs > $R/pytest_gpu.log 2>&1; tail -4 $R/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
