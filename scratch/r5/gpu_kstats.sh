#!/bin/bash
# kernel stats of one bench configuration: gpu_kstats.sh <name> [ENV=VAL ...]
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
R=$ROOT/gpurun_out/r05_kstats
mkdir -p $R
name=$1; shift
cd /tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --steps 60 --warmup 10"
env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/st_$name -- $BENCH > $R/$name.log 2>&1
echo "== $name $@"
python $ROOT/scratch/r5/kstats.py $R/st_$name 7
rm -rf $R/st_$name
