#!/bin/bash
# round 3, GPU call 2: GPU tests + config 2 bench with the lean solver stage on / off
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_gpu.log 2>&1; tail -3 gpurun_out/r03/pytest_gpu.log
for lean in 1 0; do
RP_LEAN=$lean timeout 300 python bench.py --config 2 --steps 200 --warmup 10 --no-cpu-baseline --aux-fp32 0 --host-io 0 > gpurun_out/r03/q_lean$lean.json 2> gpurun_out/r03/q_lean$lean.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r03/q_lean$lean.json").read().strip().splitlines()[-1])
l=d.get("aux",{}).get("lockstep_full_episode") or {}
r=d["roofline"]
print("lean $lean value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "seq", round(r["step_sequence_avg_ms"],3), "sol", round(r["kernel_avg_ms"],4), "envs/launch", round(r["envs_per_launch"]), "lockstep", l.get("value"), l.get("kernel_avg_ms"), d["sanity"], d.get("cpu_baseline_parity"))
PY
done
