#!/bin/bash
# instruction-cache / fetch counters of the stage kernels (separate passes, kernel-trace only)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/icache; rm -rf $R; mkdir -p $R
cd /tmp
SHORT="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --engine-only --steps 4 --warmup 1 --fingertips ${FT:-primitive}"
export RP_STREAM_SLICES=1
timeout 400 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $R/p1 -- $SHORT > $R/p1.log 2>&1
timeout 400 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/p2 -- $SHORT > $R/p2.log 2>&1
true
true
python - <<PY
import csv, glob, collections
for p in ("p1","p2","p3","p4"):
    fs = glob.glob("$R/%s/**/*counter_collection.csv" % p, recursive=True)
    if not fs: print(p, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        k = 'FUSED' if 'fused_steps' in k else 'LEAN' if 'lean' in k else 'HEAVY' if 'rp_stage_kernel<double, 1' in k else 'POS' if 'rp_stage_kernel<double, 0' in k else None
        if not k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k in acc:
        print(p, k, {c: round(v / n[(k, c)]) for c, v in acc[k].items()}, "launches", max(n[(k, c)] for c in acc[k]))
PY
rm -rf $R/p1 $R/p2 $R/p3 $R/p4
