#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r04_call5
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
bash scratch/r4/ab_trees.sh r04_call5/ab_c2 --config 2 --steps 158 --warmup 10 --fingertips hull
(cd scratch/r4/ab_old && timeout 300 python $GRAFT_REPO_ROOT/scratch/phase_prof.py 64 4096 hull > $R/phase_old.txt 2>&1)
timeout 300 python scratch/phase_prof.py 64 4096 hull > $R/phase_new.txt 2>&1
paste <(grep "cyc/mj_step" $R/phase_old.txt | sort) <(grep "cyc/mj_step" $R/phase_new.txt | sort) | head -40
grep "total cycles" $R/phase_old.txt $R/phase_new.txt
