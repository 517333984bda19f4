#!/bin/bash
# third SQ pass: the fp64 arithmetic counters + branches (short lockstep run, one stream slice)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/prof_sq3; rm -rf $R; mkdir -p $R
cd /tmp
export RP_STREAM_SLICES=1
timeout 500 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_BRANCH --output-format csv -d $R/sq3 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --stagger 0 --steps 4 --warmup 1 > $R/sq3.log 2>&1
python - <<PY
import glob, json, pandas as pd
f = sorted(glob.glob("$R/sq3/*/*counter_collection.csv"))
df = pd.read_csv(f[-1])
out = {}
for tag, pat in (("solver rp_stage_kernel<double, 1, 4, 9>", "<double, 1"), ("position rp_stage_kernel<double, 0, 0, 9>", "<double, 0")):
    x = df[df.Kernel_Name.str.contains(pat, regex=False)]
    if "<double, 0" in pat:
        big = x.groupby("Dispatch_Id").Counter_Value.sum(); x = x[x.Dispatch_Id.isin(big[big > big.max() * 0.05].index)]
    out[tag] = {k: float(v) / 4096.0 for k, v in x.groupby("Counter_Name").Counter_Value.mean().items()}
json.dump(out, open("$R/sq3.json", "w"), indent=1); print(json.dumps(out, indent=1))
PY
rm -rf $R/sq3
