#!/bin/bash
mkdir -p gpurun_out/r03
for g in 0 1; do
timeout 300 python bench.py --config 2 --steps 160 --warmup 10 --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --graph $g > gpurun_out/r03/graph$g.json 2> gpurun_out/r03/graph$g.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r03/graph$g.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("graph $g value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "seq", round(r["step_sequence_avg_ms"],3), d["config"]["hipgraph_step"])
PY
done
tail -2 gpurun_out/r03/graph1.err
