#!/bin/bash
# A/B of engine builds on one box, both fingertip colliders
mkdir -p gpurun_out/r03
for lib in "$@"; do
name=$(basename $lib .so)
for ft in hull primitive; do
RP_ENGINE_LIB=$PWD/$lib timeout 300 python bench.py --config 2 --steps ${STEPS:-120} --warmup 10 --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --fingertips $ft > gpurun_out/r03/abb_${name}_$ft.json 2> gpurun_out/r03/abb_${name}_$ft.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r03/abb_${name}_$ft.json").read().strip().splitlines()[-1])
l=d.get("aux",{}).get("lockstep_full_episode") or {}
r=d["roofline"]
print("$name $ft value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "seq", round(r["step_sequence_avg_ms"],3), "sol", round(r["kernel_avg_ms"],4), "lockstep", round(l.get("value") or 0))
PY
done
done
