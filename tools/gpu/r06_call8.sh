#!/bin/bash
# Round 6, call 8: lean-solver instruction diet A/B (dynamic VALU count and wave cycles per wave, bench).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r06_call8
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
Q="--no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --aux-rccl 0"
run() { name=$1; shift; timeout 400 env "$@" > $R/$name.json 2> $R/$name.err; python -c "
import json,sys
d=json.loads([l for l in open('$R/$name.json').read().splitlines() if l.startswith('{\"metric\"')][-1]); print('$name', round(d['value']), round(d['ms_per_step'],3), d['roofline'].get('schedule'), d['roofline'].get('kernel_avg_ms'))"; }
for rep in 1 2; do
run A_$rep python bench.py $Q --steps 316
run B_$rep RP_ENGINE_LIB=$GRAFT_REPO_ROOT/robopianist_amd/csrc/librp_engine_c.so python bench.py $Q --steps 316
done
pmc() { name=$1; shift; cd /tmp; env "$@" RP_STREAM_SLICES=1 RP_SPLIT_POS=0 timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $R/pmc_$name -- python $GRAFT_REPO_ROOT/bench.py $Q --stagger 0 --steps 4 --warmup 1 > $R/pmc_$name.log 2>&1; cd $GRAFT_REPO_ROOT; python - <<PY
import glob, pandas as pd
f = sorted(glob.glob("$R/pmc_$name/*/*counter_collection.csv"))[-1]
df = pd.read_csv(f); x = df[df.Kernel_Name.str.contains("rp_lean_solver_kernel", regex=False)]
big = x.groupby("Dispatch_Id").Counter_Value.sum(); x = x[x.Dispatch_Id.isin(big[big > big.max() * 0.05].index)]
print("$name", {k: round(v / 4096, 1) for k, v in x.groupby("Counter_Name").Counter_Value.mean().items()})
PY
rm -rf $R/pmc_$name; }
pmc A X=1
pmc B RP_ENGINE_LIB=$GRAFT_REPO_ROOT/robopianist_amd/csrc/librp_engine_c.so
