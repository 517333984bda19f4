#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "other_topologies or box_box or sensors or all_six" 2>&1 | tail -25
