#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r02c
mkdir -p $R
cd $GRAFT_REPO_ROOT
( python -m pytest tests -m gpu -x -q ) > $R/pytest_gpu.log 2>&1
tail -4 $R/pytest_gpu.log
for c in 2 3 4 5; do python bench.py --config $c --steps 150 --no-cpu-baseline --aux-fp32 0 --host-io 0 > $R/bench_c$c.json 2> $R/bench_c$c.err; done
for f in $R/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    print(sys.argv[1].split('/')[-1], "value %.0f"%j["value"], "solver_ms %.4f"%j["roofline"]["kernel_avg_ms"], "seq %.3f"%j["roofline"]["step_sequence_avg_ms"], {k:(round(v.get('value')) if isinstance(v,dict) and v.get('value') else None) for k,v in j.get("aux",{}).items()}, j["sanity"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
