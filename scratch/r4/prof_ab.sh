#!/bin/bash
# rocprofv3 kernel stats of the old tree (scratch/r4/ab_old) and the working tree, same box, same flags.
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
R=$ROOT/gpurun_out/$1; shift
mkdir -p $R
FLAGS="--no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --config 2 --steps 80 --warmup 10 --fingertips hull $@"
cd /tmp
for t in old new; do
  if [ $t = old ]; then D=$ROOT/scratch/r4/ab_old; else D=$ROOT; fi
  (cd $D && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$t -- python bench.py $FLAGS > $R/prof_$t.log 2>&1)
  f=$(find /tmp/prof_$t -name "*kernel_stats.csv" | head -1)
  cp $f $R/kernel_stats_$t.csv
  echo "== $t"; python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:9]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):6d} avg_us {float(r["AverageNs"])/1e3:9.1f} pct {float(r["Percentage"]):5.1f}')
PY
done
