#!/bin/bash
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
R=$ROOT/gpurun_out/r05_call14
rm -rf $R; mkdir -p $R
cd $ROOT
FLAGS="--no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0"
summ() {
python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    r = d["roofline"]; s = d.get("sanity", {})
    print(sys.argv[1], "value", round(d["value"]), "ms/step", round(d["ms_per_step"], 3), "seq", round(r.get("step_sequence_avg_ms") or 0, 3),
          "sol", round(r.get("kernel_avg_ms") or 0, 4), "sched", r.get("schedule"), "overflow_eps", s.get("capacity_overflow_episodes"), "warn_or", s.get("warn_flags_or"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run() { name=$1; shift; env "$@" timeout 400 python bench.py $FLAGS $EXTRA > $R/$name.json 2> $R/$name.err; summ "$name" $R/$name.json; grep "schedule choice" $R/$name.err | tail -2; }
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "split_position" > $R/pytest_split.log 2>&1; tail -3 $R/pytest_split.log
run auto RP_SCHED_DEBUG=1
run auto_b RP_SCHED_DEBUG=1
run nosplit RP_SPLIT_POS=0 RP_SCHED_DEBUG=1
EXTRA="--fingertips primitive"
run cap_auto RP_SCHED_DEBUG=1
EXTRA="--config 3"
run c3_auto RP_SCHED_DEBUG=1
EXTRA="--config 4"
run c4_auto RP_SCHED_DEBUG=1
EXTRA="--config 5"
run c5_auto RP_SCHED_DEBUG=1
EXTRA=""
RP_SPLIT_POS=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_env.py -m gpu -q -x -k "not eight_ranks" > $R/pytest_forced_split.log 2>&1; tail -3 $R/pytest_forced_split.log
