#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python scratch/cost_trace.py 2>&1 | tail -4
