#!/bin/bash
# A/B on one box: config 2 with the lean solver stage on, then off (the full-capacity build is the control:
# boxes differ by up to 1.5x in clock, so only ratios within one call mean anything)
mkdir -p gpurun_out/r03
for lean in 1 0 1; do
RP_LEAN=$lean timeout 300 python bench.py --config 2 --steps ${STEPS:-160} --warmup 10 --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 > gpurun_out/r03/ab_lean$lean.json 2> gpurun_out/r03/ab_lean$lean.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r03/ab_lean$lean.json").read().strip().splitlines()[-1])
l=d.get("aux",{}).get("lockstep_full_episode") or {}
r=d["roofline"]
print("lean $lean value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "seq", round(r["step_sequence_avg_ms"],3), "sol", round(r["kernel_avg_ms"],4), "envs/launch", round(r["envs_per_launch"]), "lockstep", round(l.get("value") or 0), l.get("kernel_avg_ms"), d["sanity"])
PY
done
