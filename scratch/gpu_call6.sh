#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python scratch/debug_deep.py 2>&1 | tail -40
python scratch/run_variants.py 2>&1 | tail -5
