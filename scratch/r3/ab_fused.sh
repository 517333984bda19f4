#!/bin/bash
# fused substeps on/off on one box, both fingertip colliders
mkdir -p gpurun_out/r03
for ft in ${FTS:-primitive hull}; do
for fu in ${FUSED:-0 1}; do
RP_FUSED=$fu timeout 300 python bench.py --config ${CFG:-2} --steps ${STEPS:-120} --warmup 10 --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --fingertips $ft $EXTRA > gpurun_out/r03/fu_${fu}_$ft.json 2> gpurun_out/r03/fu_${fu}_$ft.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03/fu_${fu}_$ft.json").read().strip().splitlines()[-1])
    r=d["roofline"]; l=d.get("aux",{}).get("lockstep_full_episode") or {}
    print("$ft fused $fu value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "seq", round(r["step_sequence_avg_ms"],3), "probe", round(r["kernel_avg_ms"],4), "units/launch", round(r["envs_per_launch"]), "lockstep", round(l.get("value") or 0), d["sanity"])
except Exception as ex:
    print("$ft fused $fu FAILED", ex); print(open("gpurun_out/r03/fu_${fu}_$ft.err").read()[-1500:])
PY
done
done
