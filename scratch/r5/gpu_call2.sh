#!/bin/bash
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
R=$ROOT/gpurun_out/r05_call2
rm -rf $R; mkdir -p $R
cd /tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --steps 60 --warmup 10"
for sl in 1 2; do for sp in 1 0; do
  echo "== slices $sl split $sp"
  RP_STREAM_SLICES=$sl RP_FUSED=0 RP_SPLIT_POS=$sp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/st_${sl}_$sp -- $BENCH > $R/st_${sl}_$sp.log 2>&1
  grep -o '"value": [0-9.]*' $R/st_${sl}_$sp.log | head -1
  python $ROOT/scratch/r5/kstats.py $R/st_${sl}_$sp 9
  rm -rf $R/st_${sl}_$sp
done; done
