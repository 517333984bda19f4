#!/bin/bash
# full GPU suite + config 2 (primitive `value`, hull aux) + config 3 after a narrow-phase change
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_np.log 2>&1; tail -2 gpurun_out/r02/pytest_np.log
for c in 2 3; do
timeout 400 python bench.py --config $c --steps 200 --warmup 10 --no-cpu-baseline --aux-fp32 0 --host-io 0 > gpurun_out/r02/np_c$c.json 2> gpurun_out/r02/np_c$c.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r02/np_c$c.json").read().strip().splitlines()[-1])
a=d.get("aux",{})
print("config $c value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "seq", round(d["roofline"]["step_sequence_avg_ms"],3), {k:round(v["value"]) for k,v in a.items()}, d["sanity"])
PY
done
