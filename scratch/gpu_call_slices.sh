#!/bin/bash
# A/B of the two-slice stepping (RP_STREAM_SLICES=1|2) + the GPU test suite with slices on.
mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu_sl.log 2>&1; tail -3 gpurun_out/r02/pytest_gpu_sl.log
for sl in 1 2; do for c in 2 3; do
RP_STREAM_SLICES=$sl timeout 300 python bench.py --config $c --steps 150 --warmup 10 --no-cpu-baseline --aux-fp32 0 --host-io 0 > gpurun_out/r02/sl${sl}_c$c.json 2> gpurun_out/r02/sl${sl}_c$c.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r02/sl${sl}_c$c.json").read().strip().splitlines()[-1])
print("slices $sl config $c value", round(d["value"]), "ms/step", d["ms_per_step"], "lockstep", d.get("aux",{}).get("lockstep_full_episode"))
PY
done; done
