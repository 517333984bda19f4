#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
cd $GRAFT_REPO_ROOT
( python -m pytest tests -m gpu -x -q ) > gpurun_out/r02/pytest_gpu5.log 2>&1
tail -25 gpurun_out/r02/pytest_gpu5.log
for cfg in 2 3 4 5; do
  python bench.py --config $cfg --steps 150 --warmup 10 --no-cpu-baseline --aux-fp32 0 --host-io 0 > gpurun_out/r02/g5_c${cfg}.json 2> gpurun_out/r02/g5_c${cfg}.err
  python - gpurun_out/r02/g5_c${cfg}.json <<'PY'
import json,sys
try:
    j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    print(sys.argv[1], "value %.0f"%j["value"], "ms/step %.3f"%j["ms_per_step"], "solver_ms %.4f"%j["roofline"]["kernel_avg_ms"], "seq_ms %.3f"%j["roofline"]["step_sequence_avg_ms"], j["sanity"])
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
python bench.py --config 2 --stagger 0 --steps 158 --warmup 10 --no-cpu-baseline --aux-fp32 0 --host-io 0 | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lockstep', j['value'], j['roofline']['kernel_avg_ms'], j['roofline']['step_sequence_avg_ms'])"
RP_COST_ORDER=0 python bench.py --config 2 --stagger 0 --steps 158 --warmup 10 --no-cpu-baseline --aux-fp32 0 --host-io 0 | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lockstep co0', j['value'], j['roofline']['kernel_avg_ms'], j['roofline']['step_sequence_avg_ms'])"
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02/prof5 -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --aux-fp32 0 --host-io 0 > $GRAFT_REPO_ROOT/gpurun_out/r02/prof5_bench.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/r02/prof5 -name "*kernel_stats.csv" | head -1 | xargs -I{} head -8 {}
