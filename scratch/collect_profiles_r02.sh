#!/bin/bash
# Round-2 profile collection (GPU box): bench line, kernel trace + stats, PMC passes (separate runs).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/prof_r02
mkdir -p $R
cd $GRAFT_REPO_ROOT
python bench.py > $R/bench_plain.json 2> $R/bench_plain.err
for c in 3 4 5; do python bench.py --config $c --steps 150 > $R/bench_c$c.json 2> $R/bench_c$c.err; done
cd /tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --aux-fp32 0 --host-io 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats -- $BENCH --steps 158 --warmup 20 > $R/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/fetch -- $BENCH --steps 20 --warmup 5 > $R/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/write -- $BENCH --steps 20 --warmup 5 > $R/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/sq1 -- $BENCH --steps 10 --warmup 3 > $R/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d $R/sq2 -- $BENCH --steps 10 --warmup 3 > $R/sq2.log 2>&1
cd $GRAFT_REPO_ROOT
python scratch/summarize_profiles_r02.py $R 2>&1 | tail -60
# only the summaries travel back (the raw traces are > 64 MiB)
for d in stats fetch write sq1 sq2; do rm -rf $R/$d; done
du -sh $R
