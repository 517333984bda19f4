#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/tl; rm -rf $R; mkdir -p $R
cd /tmp
for ft in ${FTS:-primitive hull}; do
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$ft -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --fingertips $ft --steps 60 --warmup 10 $EXTRA > $R/$ft.log 2>&1
echo == $ft; tail -1 $R/$ft.log | cut -c1-120
python $GRAFT_REPO_ROOT/scratch/r3/timeline.py $R/$ft $SEG
rm -rf $R/$ft
done
