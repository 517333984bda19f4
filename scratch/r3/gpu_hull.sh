#!/bin/bash
mkdir -p gpurun_out/r03
RP_SKIP_SELF_CHECK=1 timeout 300 python scratch/phase_prof.py 64 4096 hull > gpurun_out/r03/phase_prof_hull.log 2>&1
grep -v "^W2026\|amdgpu.ids" gpurun_out/r03/phase_prof_hull.log | head -16
for ft in hull primitive; do
timeout 300 python bench.py --config 2 --steps 120 --warmup 10 --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --fingertips $ft > gpurun_out/r03/ft_$ft.json 2> gpurun_out/r03/ft_$ft.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r03/ft_$ft.json").read().strip().splitlines()[-1])
l=d.get("aux",{}).get("lockstep_full_episode") or {}
r=d["roofline"]
print("$ft value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "seq", round(r["step_sequence_avg_ms"],3), "sol", round(r["kernel_avg_ms"],4), "lockstep", round(l.get("value") or 0), d["sanity"])
PY
done
