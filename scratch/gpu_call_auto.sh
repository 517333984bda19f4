#!/bin/bash
# configs 2/3 with forced 1 / 2 / 4 stream slices.
mkdir -p gpurun_out/r02
for sl in 4 2 1; do for c in 2 3; do
RP_STREAM_SLICES=$sl timeout 300 python bench.py --config $c --steps 200 --warmup 10 --no-cpu-baseline --aux-fp32 0 --host-io 0 > gpurun_out/r02/au${sl}_c$c.json 2> gpurun_out/r02/au${sl}_c$c.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r02/au${sl}_c$c.json").read().strip().splitlines()[-1])
l=d.get("aux",{}).get("lockstep_full_episode") or {}
r=d["roofline"]
print("slices $sl config $c value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "seq", round(r["step_sequence_avg_ms"],3), "sol", round(r["kernel_avg_ms"],4), "envs/launch", r["envs_per_launch"], "frac", round(r["frac"],4), "lockstep", l.get("value"), l.get("envs_per_launch"))
PY
done; done
