#!/bin/bash
# Round 6: SQ instruction counters of the SPLIT position stage's kernels (front / pooled narrow / back) next to the lean
# solver stage, one slice so that launches do not overlap.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r06_call13; rm -rf $R; mkdir -p $R
cd /tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --aux-rccl 0 --steps 8 --warmup 160"
export RP_STREAM_SLICES=1 RP_SPLIT_POS=1
timeout 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/a -- $BENCH > $R/a.log 2>&1
timeout 500 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES --output-format csv -d $R/b -- $BENCH > $R/b.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, os, json
R = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_call13"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for d in ("a", "b"):
    for f in glob.glob(R + "/" + d + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        # the staggered steady state: the last 8 control steps' launches = the last ~80 launches per kernel
        per = collections.defaultdict(list)
        for row in rows: per[(row["Kernel_Name"].split("(")[0][:48], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in per.items():
            v = v[-80:] if len(v) > 80 else v
            acc[k][c] = sum(v) / len(v); n[k] = len(v)
out = {k: dict(v, launches_sampled=n[k]) for k, v in acc.items() if "rp_" in k and "float" not in k}
json.dump(out, open(R + "/split_sq.json", "w"), indent=1)
for k, v in out.items(): print(k, {c: round(x) for c, x in v.items()})
PY
rm -rf $R/a $R/b
