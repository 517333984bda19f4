#!/bin/bash
# A/B of engine builds on one box (both fingertip colliders) + the per-phase cycle profile of each
bash scratch/r3/ab_both.sh "$@"
for l in "$@"; do echo == $l; RP_ENGINE_LIB=$PWD/$l python scratch/phase_prof.py 64 4096 2>&1 | tail -26 | head -${NPH:-16}; done
