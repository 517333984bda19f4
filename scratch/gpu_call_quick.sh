#!/bin/bash
# quick: whole GPU parity file + config 2 / 3 bench
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu_q.log 2>&1; tail -3 gpurun_out/r02/pytest_gpu_q.log
for c in 2 3; do
timeout 300 python bench.py --config $c --steps 200 --warmup 10 --no-cpu-baseline --aux-fp32 0 --host-io 0 > gpurun_out/r02/q_c$c.json 2> gpurun_out/r02/q_c$c.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r02/q_c$c.json").read().strip().splitlines()[-1])
l=d.get("aux",{}).get("lockstep_full_episode") or {}
r=d["roofline"]
print("config $c value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "seq", round(r["step_sequence_avg_ms"],3), "sol", round(r["kernel_avg_ms"],4), "envs/launch", round(r["envs_per_launch"]), "lockstep", l.get("value"), l.get("kernel_avg_ms"), d["sanity"])
PY
done
