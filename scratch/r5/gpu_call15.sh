#!/bin/bash
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
R=$ROOT/gpurun_out/r05_call15
rm -rf $R; mkdir -p $R
cd $ROOT
timeout 2400 python -m pytest tests -m gpu -q -x > $R/pytest_gpu.log 2>&1; tail -5 $R/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $R/smoke.log 2>&1; tail -1 $R/smoke.log
timeout 900 python bench.py > $R/bench_plain.json 2> $R/bench_plain.err; tail -c 600 $R/bench_plain.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_call15/bench_plain.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], d["roofline"].get("schedule"))
print("config.solve_stats", d["config"].get("solve_stats_last_step"))
print("standin", {k: v for k, v in d["config"].get("standin_contacts", {}).items() if k != "note"})
p = d.get("cpu_baseline_parity", {})
print("parity", {k: p.get(k) for k in ("max_rel_qpos_error_1000_mj_steps", "first_mj_step_above_1e-6", "teacher_forced_worst_rel_dv", "teacher_forced_mj_steps", "teacher_forced_contact_count_mismatches")})
print("control", {k: v for k, v in p.get("chaos_control", {}).items() if k != "note"})
print("aux", {k: (v.get("value") if isinstance(v, dict) else v) for k, v in d.get("aux", {}).items()})
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"))
PY
