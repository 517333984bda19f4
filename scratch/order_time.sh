#!/bin/bash
# average duration of rp_order_kernel (and the stage kernels) in a short one-slice lockstep run
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/ordt; rm -rf $R; mkdir -p $R
cd /tmp
RP_STREAM_SLICES=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/st -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --aux-fp32 0 --host-io 0 --stagger 0 --steps 30 --warmup 5 > $R/log 2>&1
python - <<PY
import glob, pandas as pd
f = sorted(glob.glob("$R/st/*/*kernel_stats.csv"))[-1]
d = pd.read_csv(f); d["Name"] = d["Name"].str.slice(0, 60)
print(d[["Name", "Calls", "AverageNs", "Percentage"]].head(8).to_string())
PY
rm -rf $R/st
