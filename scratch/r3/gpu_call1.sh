#!/bin/bash
# round 3, GPU call 1: problem-size statistics + per-phase profile of the current solver
mkdir -p gpurun_out
export RP_SKIP_SELF_CHECK=1
timeout 900 python scratch/r3/stats.py > gpurun_out/r3_stats.log 2>&1
timeout 300 python scratch/phase_prof.py 64 4096 > gpurun_out/r3_phase_prof.log 2>&1
tail -40 gpurun_out/r3_phase_prof.log
