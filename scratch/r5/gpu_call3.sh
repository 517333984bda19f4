#!/bin/bash
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
R=$ROOT/gpurun_out/r05_call3
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
for lib in "" "$ROOT/scratch/r5/lib/librp_engine_fpon.so"; do
  echo "=== lib: ${lib:-default}"
  export RP_ENGINE_LIB=$lib; [ -z "$lib" ] && unset RP_ENGINE_LIB
  timeout 300 python scratch/r5/ws_split_vs_whole.py 150 hull replay 300 2>&1 | tail -1
  timeout 300 python scratch/r5/ws_split_vs_whole.py 150 cap replay 300 2>&1 | tail -1
  for rep in 1; do
    RP_SPLIT_POS=0 timeout 400 python bench.py $FLAGS > $R/whole_$rep.json 2> $R/whole_$rep.err; summ "whole#$rep" $R/whole_$rep.json
    RP_SPLIT_POS=1 timeout 400 python bench.py $FLAGS > $R/split_$rep.json 2> $R/split_$rep.err; summ "split#$rep" $R/split_$rep.json
    RP_SPLIT_POS=1 RP_STREAM_SLICES=1 RP_FUSED=0 timeout 400 python bench.py $FLAGS > $R/split1_$rep.json 2> $R/split1_$rep.err; summ "split 1 slice#$rep" $R/split1_$rep.json
    RP_SPLIT_POS=0 RP_STREAM_SLICES=1 RP_FUSED=0 timeout 400 python bench.py $FLAGS > $R/whole1_$rep.json 2> $R/whole1_$rep.err; summ "whole 1 slice#$rep" $R/whole1_$rep.json
  done
done
