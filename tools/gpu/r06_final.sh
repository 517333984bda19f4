#!/bin/bash
# Round 6: last check of the committed tree: GPU tests, smoke, the default bench line.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r06_final; rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > $R/pytest_gpu.log 2>&1; tail -2 $R/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $R/smoke.log 2>&1; tail -2 $R/smoke.log
timeout 1200 python bench.py > $R/bench.json 2> $R/bench.err; python -c "
import json; d=json.loads(open('$R/bench.json').read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline_parity']['meets_bar'], d['cpu_baseline_parity']['max_rel_qpos_error_whole_stream'])"
