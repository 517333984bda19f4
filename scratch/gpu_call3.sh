#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
cd $GRAFT_REPO_ROOT
( python -m pytest tests -m gpu -x -q ) > gpurun_out/r02/pytest_gpu3.log 2>&1
tail -25 gpurun_out/r02/pytest_gpu3.log
python scratch/cost_hist.py 2>&1 | tail -8
for cfg in 2 3 4 5; do
  python bench.py --config $cfg --steps 150 --warmup 10 --no-cpu-baseline --aux-fp32 0 --host-io 0 > gpurun_out/r02/g3_c${cfg}.json 2> gpurun_out/r02/g3_c${cfg}.err
  python - gpurun_out/r02/g3_c${cfg}.json <<'PY'
import json,sys
try:
    j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    print(sys.argv[1], "value %.0f"%j["value"], "ms/step %.3f"%j["ms_per_step"], "solver_ms %.4f"%j["roofline"]["kernel_avg_ms"], "seq_ms %.3f"%j["roofline"]["step_sequence_avg_ms"], j["sanity"])
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
python bench.py --config 2 --stagger 0 --steps 158 --warmup 10 --aux-fp32 0 --host-io 0 > gpurun_out/r02/g3_c2_lockstep.json 2>gpurun_out/r02/g3_c2_lockstep.err
python -c "
import json; j=json.loads([l for l in open('gpurun_out/r02/g3_c2_lockstep.json') if l.startswith('{')][0]); print('lockstep', j['value'], j['roofline']['kernel_avg_ms'], j['cpu_baseline_parity'])"
