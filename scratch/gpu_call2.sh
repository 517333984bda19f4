#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
cd $GRAFT_REPO_ROOT
( python -m pytest tests -m gpu -x -q -k "sensor or torque or plain_c or teacher_forced_fp64_random" ) > gpurun_out/r02/pytest_gpu2.log 2>&1
tail -15 gpurun_out/r02/pytest_gpu2.log
for co in 0 1; do
  for cfg in 2 3 4 5; do
    RP_COST_ORDER=$co python bench.py --config $cfg --steps 150 --warmup 10 --no-cpu-baseline --aux-fp32 0 --host-io 0 > gpurun_out/r02/co${co}_c${cfg}.json 2> gpurun_out/r02/co${co}_c${cfg}.err
    python - gpurun_out/r02/co${co}_c${cfg}.json <<'PY'
import json,sys
try:
    j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    print(sys.argv[1], "value %.0f"%j["value"], "ms/step %.3f"%j["ms_per_step"], "solver_ms %.4f"%j["roofline"]["kernel_avg_ms"], "seq_ms %.3f"%j["roofline"]["step_sequence_avg_ms"], j["sanity"])
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
  done
done
RP_COST_ORDER=1 python bench.py --config 2 --stagger 0 --steps 158 --warmup 10 --no-cpu-baseline --aux-fp32 0 --host-io 0 | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lockstep co1', j['value'], j['roofline']['kernel_avg_ms'])"
RP_COST_ORDER=0 python bench.py --config 2 --stagger 0 --steps 158 --warmup 10 --no-cpu-baseline --aux-fp32 0 --host-io 0 | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lockstep co0', j['value'], j['roofline']['kernel_avg_ms'])"
