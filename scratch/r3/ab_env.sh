#!/bin/bash
# A/B of one environment variable on one box: ab_env.sh NAME "v1 v2 ..." ; FTS, CFG, STEPS, EXTRA as in ab_fused.sh
mkdir -p gpurun_out/r03
name=$1; vals=$2
for ft in ${FTS:-primitive hull}; do
for v in $vals; do
env $name=$v timeout 300 python bench.py --config ${CFG:-2} --steps ${STEPS:-120} --warmup 10 --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --fingertips $ft $EXTRA > gpurun_out/r03/env_${v}_$ft.json 2> gpurun_out/r03/env_${v}_$ft.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03/env_${v}_$ft.json").read().strip().splitlines()[-1])
    r=d["roofline"]; l=d.get("aux",{}).get("lockstep_full_episode") or {}
    print("$ft $name=$v value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "seq", round(r["step_sequence_avg_ms"],3), "probe", round(r["kernel_avg_ms"],4), "units", round(r["envs_per_launch"]), "lockstep", round(l.get("value") or 0))
except Exception as ex:
    print("$ft $name=$v FAILED", ex); print(open("gpurun_out/r03/env_${v}_$ft.err").read()[-800:])
PY
done
done
