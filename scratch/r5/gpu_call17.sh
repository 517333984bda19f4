#!/bin/bash
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
R=$ROOT/gpurun_out/r05_call17
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
          "sol", round(r.get("kernel_avg_ms") or 0, 4), "sched", r.get("schedule")[:30], "overflow_eps", s.get("capacity_overflow_episodes"), "warn_or", s.get("warn_flags_or"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run() { name=$1; shift; env "$@" timeout 400 python bench.py $FLAGS $EXTRA > $R/$name.json 2> $R/$name.err; summ "$name" $R/$name.json; grep "schedule choice" $R/$name.err | head -1; }
run auto RP_SCHED_DEBUG=1
run auto_b RP_SCHED_DEBUG=1
run whole2 RP_SPLIT_POS=0 RP_STREAM_SLICES=2 RP_FUSED=0
EXTRA="--fingertips primitive"
run cap_auto RP_SCHED_DEBUG=1
EXTRA=""
RP_SPLIT_POS=1 RP_STREAM_SLICES=1 RP_FUSED=0 timeout 300 python scratch/phase_prof.py 64 4096 hull 2>&1 | grep "MPR\|narrow geom\|list\|total"
RP_SPLIT_POS=0 RP_STREAM_SLICES=1 RP_FUSED=0 timeout 300 python scratch/phase_prof.py 64 4096 hull 2>&1 | grep "MPR\|narrow geom\|total"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hull or split or replay" 2>&1 | tail -2
