#!/bin/bash
# Builds librp_engine variants with different backend options (perf experiment; the .so files
# travel to the GPU box, scratch/run_variants.py times each).  Usage: scratch/build_variants.sh
cd "$(dirname "$0")/.."
C=robopianist_amd/csrc
OUT=$C/variants
mkdir -p $OUT
build() {  # name, extra flags...
  name=$1; shift
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value "$@" $C/rp_engine.hip $C/rp_task.hip -o $OUT/librp_engine_$name.so &
}
build base
build maxilp -mllvm -amdgpu-sched-strategy=max-ilp
build maxmem -mllvm -amdgpu-sched-strategy=max-memory-clause
build trackers -mllvm -amdgpu-use-amdgpu-trackers=1
build nopostsched -mllvm -enable-post-misched=0
build bias0 -mllvm -amdgpu-schedule-metric-bias=0
build preallocsgpr -mllvm -amdgpu-prealloc-sgpr-spill-vgprs=1
build O2 -O2
wait
ls -la $OUT
