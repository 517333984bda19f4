#!/bin/bash
mkdir -p gpurun_out/r03
for c in 3 4 5; do
for ft in primitive hull; do
timeout 400 python bench.py --config $c --steps 150 --warmup 10 --no-cpu-baseline --fingertips $ft > gpurun_out/r03/cfg${c}_$ft.json 2> gpurun_out/r03/cfg${c}_$ft.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r03/cfg${c}_$ft.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("config $c $ft value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "sol", round(r["kernel_avg_ms"],4), "envs/launch", round(r["envs_per_launch"]), d["sanity"])
PY
done
done
