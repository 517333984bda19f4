#!/bin/bash
# GPU call 1 of round 2: tests, the four bench configs, backend-flag variants, kernel trace.
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
cd $GRAFT_REPO_ROOT
( time python -m pytest tests -m gpu -x -q --durations=8 ) > gpurun_out/r02/pytest_gpu.log 2>&1
tail -5 gpurun_out/r02/pytest_gpu.log
python bench.py > gpurun_out/r02/bench_c2.json 2> gpurun_out/r02/bench_c2.err
python bench.py --config 3 --steps 150 > gpurun_out/r02/bench_c3.json 2> gpurun_out/r02/bench_c3.err
python bench.py --config 4 --steps 150 > gpurun_out/r02/bench_c4.json 2> gpurun_out/r02/bench_c4.err
python bench.py --config 5 --steps 150 > gpurun_out/r02/bench_c5.json 2> gpurun_out/r02/bench_c5.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --aux-fp32 0 --host-io 0 > gpurun_out/r02/bench_c2_driver_window.json 2>&1
python scratch/run_variants.py > gpurun_out/r02/variants.log 2>&1
cat gpurun_out/r02/variants.log
for f in gpurun_out/r02/bench_c*.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    print(sys.argv[1], "value %.0f"%j["value"], "ms/step %.3f"%j["ms_per_step"], "solver_ms %.4f"%j["roofline"]["kernel_avg_ms"], j["sanity"], j.get("cpu_baseline",{}).get("value"), (j.get("cpu_baseline_parity") or {}).get("max_rel_qpos_error_1000_mj_steps"), (j.get("cpu_baseline_parity") or {}).get("teacher_forced_worst_rel_dv_300_mj_steps"))
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:] if sys.argv[1].endswith(".json") else "")
PY
done
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 158 --warmup 10 --no-cpu-baseline --aux-fp32 0 --host-io 0 > $GRAFT_REPO_ROOT/gpurun_out/r02/prof_bench.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/r02/prof -name "*kernel_stats.csv" | head -2 | xargs -I{} head -12 {}
