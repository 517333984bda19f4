#!/bin/bash
# A/B inside one gpurun call: the tree in scratch/r4/ab_old (a copy of an earlier commit, its own library built) against
# the working tree.  Usage: ab_trees.sh <out-dir-name> [bench flags...]
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
R=$ROOT/gpurun_out/$1; shift
mkdir -p $R
FLAGS="--no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 $@"
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
for rep in 1 2; do
  (cd $ROOT/scratch/r4/ab_old && timeout 400 python bench.py $FLAGS > $R/old_$rep.json 2> $R/old_$rep.err); summ "old#$rep" $R/old_$rep.json
  (cd $ROOT && timeout 400 python bench.py $FLAGS --aux-large-hulls 0 > $R/new_$rep.json 2> $R/new_$rep.err); summ "new#$rep" $R/new_$rep.json
done
