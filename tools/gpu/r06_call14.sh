#!/bin/bash
# Round 6: is the LDS (one per CU, eight solver waves behind it) what co-resident waves wait for?  Bank-conflict and
# LDS-busy counters of the lean solver stage and the split position stage's kernels, one slice.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r06_call14; rm -rf $R; mkdir -p $R
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_LDS[A-Z_0-9]*\|SQ_INSTS_LDS[A-Z_0-9]*\|TCP_[A-Z_0-9]*LATENCY[A-Z_0-9]*\|SQ_INST_LEVEL[A-Z_0-9]*\|SQ_LEVEL_WAVES\|SQ_ACCUM_PREV[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $R/counters.txt
BENCH="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --aux-rccl 0 --steps 8 --warmup 160"
export RP_STREAM_SLICES=1 RP_SPLIT_POS=1
timeout 500 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d $R/a -- $BENCH > $R/a.log 2>&1
timeout 500 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $R/b -- $BENCH > $R/b.log 2>&1
timeout 500 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU --output-format csv -d $R/c -- $BENCH > $R/c.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, os, json
R = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_call14"
acc = collections.defaultdict(dict); n = {}
for d in ("a", "b", "c"):
    for f in glob.glob(R + "/" + d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(list)
        for row in csv.DictReader(open(f)): per[(row["Kernel_Name"].split("(")[0][:48], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in per.items():
            v = v[-80:] if len(v) > 80 else v
            acc[k][c] = sum(v) / len(v); n[k] = len(v)
out = {k: dict(v, launches_sampled=n[k]) for k, v in acc.items() if "rp_" in k and "float" not in k and "reset" not in k}
json.dump(out, open(R + "/lds_sq.json", "w"), indent=1)
for k, v in out.items(): print(k, {c: round(x) for c, x in v.items()})
PY
tail -3 $R/a.log | cut -c1-300
rm -rf $R/a $R/b $R/c; cat $R/counters.txt
