#!/bin/bash
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
R=$ROOT/gpurun_out/r05_call4
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
timeout 300 python scratch/r5/ws_split_vs_whole.py 150 hull replay 300 2>&1 | tail -1
RP_SPLIT_POS=0 timeout 400 python bench.py $FLAGS > $R/whole.json 2> $R/whole.err; summ "whole" $R/whole.json
RP_SPLIT_POS=1 timeout 400 python bench.py $FLAGS > $R/split.json 2> $R/split.err; summ "split" $R/split.json
RP_SPLIT_POS=1 RP_NARROW_GRID=2048 timeout 400 python bench.py $FLAGS > $R/split_g2048.json 2> $R/split.err; summ "split grid 2048" $R/split_g2048.json
RP_SPLIT_POS=1 RP_NARROW_GRID=512 timeout 400 python bench.py $FLAGS > $R/split_g512.json 2> $R/split.err; summ "split grid 512" $R/split_g512.json
RP_SPLIT_POS=1 RP_STREAM_SLICES=1 RP_FUSED=0 timeout 400 python bench.py $FLAGS > $R/split1.json 2> $R/split1.err; summ "split 1 slice" $R/split1.json
RP_SPLIT_POS=0 RP_STREAM_SLICES=1 RP_FUSED=0 timeout 400 python bench.py $FLAGS > $R/whole1.json 2> $R/whole1.err; summ "whole 1 slice" $R/whole1.json
cd /tmp
BENCH="python $ROOT/bench.py $FLAGS --steps 60 --warmup 10"
for sl in 1 2; do for sp in 1; do
  echo "== slices $sl split $sp"
  RP_STREAM_SLICES=$sl RP_FUSED=0 RP_SPLIT_POS=$sp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/st_${sl}_$sp -- $BENCH > $R/st_${sl}_$sp.log 2>&1
  grep -o '"value": [0-9.]*' $R/st_${sl}_$sp.log | head -1
  python $ROOT/scratch/r5/kstats.py $R/st_${sl}_$sp 7
  [ $sl = 2 ] && python $ROOT/scratch/r4/timeline.py $R/st_${sl}_$sp 60 | sed -n 1,8p\;18,80p
  rm -rf $R/st_${sl}_$sp
done; done
