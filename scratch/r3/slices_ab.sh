#!/bin/bash
# stream-slice modes on one box: 0 = automatic, 1, 2, 4
mkdir -p gpurun_out/r03
for ft in ${FTS:-primitive hull}; do
for sl in ${SLICES:-0 1 2 4}; do
RP_STREAM_SLICES=$sl timeout 300 python bench.py --config 2 --steps ${STEPS:-120} --warmup 10 --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --fingertips $ft $EXTRA > gpurun_out/r03/sl_${sl}_$ft.json 2> gpurun_out/r03/sl_${sl}_$ft.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r03/sl_${sl}_$ft.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("$ft slices $sl value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "seq", round(r["step_sequence_avg_ms"],3), "sol", round(r["kernel_avg_ms"],4), "envs/launch", round(r["envs_per_launch"]))
PY
done
done
