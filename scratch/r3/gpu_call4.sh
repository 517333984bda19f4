#!/bin/bash
# round 3: full GPU suite + smoke + the default bench line (value = hull fingertips)
mkdir -p gpurun_out/r03
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_gpu.log 2>&1; tail -3 gpurun_out/r03/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r03/bench_default.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value","value_hull_fingertips","value_primitive_fingertips","ms_per_step")})
print("roofline", {k: d["roofline"].get(k) for k in ("achieved","frac","kernel_avg_ms","envs_per_launch","step_sequence_avg_ms","valu")})
print("cpu", d.get("cpu_baseline"))
for k in ("cpu_baseline_parity","cpu_baseline_parity_primitive_fingertips"):
    p=d.get(k) or {}
    print(k, {q: p.get(q) for q in ("max_rel_qpos_error_1000_mj_steps","teacher_forced_worst_rel_dv_300_mj_steps","teacher_forced_contact_count_mismatches","teacher_forced_max_contacts")})
print("aux", {k: (v.get("value") if isinstance(v, dict) else v) for k, v in (d.get("aux") or {}).items()})
print("sanity", d["sanity"])
PY
