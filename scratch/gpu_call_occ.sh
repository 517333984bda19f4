#!/bin/bash
# fp64 solver stage at 2 waves/SIMD (spilling build, reduced caps) vs the default build, lockstep config 2.
mkdir -p gpurun_out/r02
for v in default w1_b w2_b w2_a; do
  if [ $v = default ]; then unset RP_ENGINE_LIB; else export RP_ENGINE_LIB=$PWD/scratch/alt/librp_$v.so; fi
  RP_STREAM_SLICES=1 timeout 300 python bench.py --config 2 --stagger 0 --steps 158 --warmup 5 --no-cpu-baseline --aux-fp32 0 --host-io 0 > gpurun_out/r02/occ_$v.json 2> gpurun_out/r02/occ_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r02/occ_$v.json").read().strip().splitlines()[-1])
print("$v value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "solver_ms", d["roofline"].get("kernel_avg_ms"), d.get("health"))
PY
done
