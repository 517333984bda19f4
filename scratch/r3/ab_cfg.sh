#!/bin/bash
# usage: ab_cfg.sh CONFIG lib1 lib2 ... ; env CO (RP_COST_ORDER values, default "1")
mkdir -p gpurun_out/r03
cfg=$1; shift
for lib in "$@"; do
name=$(basename $lib .so)
for co in ${CO:-1}; do
RP_COST_ORDER=$co RP_ENGINE_LIB=$PWD/$lib timeout 300 python bench.py --config $cfg --steps ${STEPS:-120} --warmup 10 --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 > gpurun_out/r03/abc_${name}.json 2> gpurun_out/r03/abc_${name}.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r03/abc_${name}.json").read().strip().splitlines()[-1])
l=d.get("aux",{}).get("lockstep_full_episode") or {}
r=d["roofline"]
print("config $cfg $name cost_order=$co value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "seq", round(r["step_sequence_avg_ms"],3), "sol", round(r["kernel_avg_ms"],4), "lockstep", round(l.get("value") or 0), d["sanity"])
PY
done
done
