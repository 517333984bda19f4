#!/bin/bash
# hull-fingertip parity tests + config 2 with hull fingertips as `value`
mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_oracle.py -m gpu -x -q -k "hull or mesh or sensors or box" > gpurun_out/r02/pytest_hull.log 2>&1; tail -2 gpurun_out/r02/pytest_hull.log
timeout 300 python bench.py --fingertips hull --aux-fingertips 0 --steps 158 --warmup 10 --no-cpu-baseline --aux-fp32 0 --host-io 0 > gpurun_out/r02/hullv.json 2> gpurun_out/r02/hullv.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r02/hullv.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("hull value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "seq", round(r["step_sequence_avg_ms"],3), "sol", round(r["kernel_avg_ms"],4), d["sanity"])
PY
