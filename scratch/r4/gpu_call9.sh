#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r04_call9
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
bash scratch/r4/ab_trees.sh r04_call9/ab_c2 --config 2 --steps 158 --warmup 10 --fingertips hull
for n in 50 100 200 300; do
  timeout 400 python bench.py --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --config 2 --steps 40 --warmup 5 --aux-large-hulls $n > $R/hulls_$n.json 2> $R/hulls_$n.err
  python - $R/hulls_$n.json $n <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); a = d["aux"]["large_hulls"]
    print("large hulls", sys.argv[2], "vertices: value", round(a["value"]), "seq ms", round(a["step_sequence_avg_ms"], 3), "warn", a["sanity"]["warn_flags_or"], "| 26-vertex fingertips only:", round(d["value"]))
except Exception as e:
    print("large hulls", sys.argv[2], "FAILED", e)
PY
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_bench_cli.py -m gpu -x -q -s -k "large_hull or eight_ranks or hull_fingertips" > $R/pytest_sel.log 2>&1; grep "8 ranks\|passed\|failed\|large hulls" $R/pytest_sel.log | tail -6
