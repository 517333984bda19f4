#!/bin/bash
# Round 6: instruction-cache counters of the step's kernels (is the lean solver's 56 KB Newton loop thrashing the 64 KB
# instruction cache two CUs share, alone and next to the other slices' kernels?)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r06_call12; rm -rf $R; mkdir -p $R
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_WAIT[A-Z_0-9]*\|SQ_INST_LEVEL[A-Z_0-9]*\|SQ_BUSY[A-Z_0-9]*\|SQ_ACTIVE_INST[A-Z_0-9]*\|SQ_INST_CYCLES[A-Z_0-9]*\|SQ_THREAD_CYCLES[A-Z_0-9]*" | sort -u > $R/counters.txt
BENCH="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --aux-rccl 0 --stagger 0 --steps 4 --warmup 1"
for mode in default 1slice; do
  if [ $mode = 1slice ]; then export RP_STREAM_SLICES=1 RP_SPLIT_POS=0; fi
  timeout 500 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $R/ic_$mode -- $BENCH > $R/ic_$mode.log 2>&1
  timeout 500 rocprofv3 --pmc SQ_IFETCH SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $R/if_$mode -- $BENCH > $R/if_$mode.log 2>&1
done
unset RP_STREAM_SLICES RP_SPLIT_POS
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, os
R = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_call12"
for d in sorted(glob.glob(R + "/i[cf]_*")):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0][:60]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])] += 1
    print(os.path.basename(d))
    for k in acc:
        if "rp_" not in k: continue
        print("  ", k, {c: round(v / max(n[(k, c)], 1)) for c, v in acc[k].items()}, "launches", max(n[(k, c)] for c in acc[k]))
PY
for d in ic_default ic_1slice if_default if_1slice; do rm -rf $R/$d; done
cat $R/counters.txt | tr '\n' ' '
