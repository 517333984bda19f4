#!/bin/bash
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
R=$ROOT/gpurun_out/r05_call6
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
run() { name=$1; shift; env "$@" timeout 400 python bench.py $FLAGS $EXTRA > $R/$name.json 2> $R/$name.err; summ "$name" $R/$name.json; }
timeout 300 python scratch/r5/ws_split_vs_whole.py 150 hull replay 300 2>&1 | tail -1
RP_FUSE_LF=1 timeout 300 python scratch/r5/ws_split_vs_whole.py 150 hull replay 300 2>&1 | tail -1
run whole RP_SPLIT_POS=0
run split RP_SPLIT_POS=1
run split_fuse RP_SPLIT_POS=1 RP_FUSE_LF=1
run split_2sl RP_SPLIT_POS=1 RP_STREAM_SLICES=2 RP_FUSED=0
run split_fuse_2sl RP_SPLIT_POS=1 RP_FUSE_LF=1 RP_STREAM_SLICES=2 RP_FUSED=0
run split_4sl RP_SPLIT_POS=1 RP_STREAM_SLICES=4 RP_FUSED=0
run split_fuse_4sl RP_SPLIT_POS=1 RP_FUSE_LF=1 RP_STREAM_SLICES=4 RP_FUSED=0
run whole_4sl RP_SPLIT_POS=0 RP_STREAM_SLICES=4 RP_FUSED=0
run split_1sl RP_SPLIT_POS=1 RP_STREAM_SLICES=1 RP_FUSED=0
run split_fuse_1sl RP_SPLIT_POS=1 RP_FUSE_LF=1 RP_STREAM_SLICES=1 RP_FUSED=0
run whole_1sl RP_SPLIT_POS=0 RP_STREAM_SLICES=1 RP_FUSED=0
EXTRA="--config 3"
run c3_whole RP_SPLIT_POS=0
run c3_split RP_SPLIT_POS=1
run c3_split_fuse RP_SPLIT_POS=1 RP_FUSE_LF=1
EXTRA=""
echo "=== phase profile, split"
RP_SPLIT_POS=1 RP_STREAM_SLICES=1 RP_FUSED=0 timeout 300 python scratch/phase_prof.py 64 4096 hull 2>&1 | grep "candidate gen\|trace\|geom centres\|narrow geometry\|drain\|total"
