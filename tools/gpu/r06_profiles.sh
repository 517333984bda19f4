#!/bin/bash
# Round 6: everything behind profiles/r06_*: GPU tests + smoke + bench lines (configs 2..5, 2048 envs) + rocprofv3 kernel
# trace + the PMC passes (separate runs, per the guide).  Raw traces exceed the copy-back limit: reduced on the box
# (tools/gpu/summarize_profiles.py); the summaries are then copied into profiles/ by hand.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/prof_r06
rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
if [ -z "$SKIP_TESTS" ]; then
timeout 1500 python -m pytest tests -m gpu -q > $R/pytest_gpu.log 2>&1; tail -2 $R/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $R/smoke.log 2>&1; tail -1 $R/smoke.log
fi
timeout 1200 python bench.py > $R/bench_plain.json 2> $R/bench_plain.err
if [ -z "$SKIP_CONFIGS" ]; then
for c in 3 4 5; do timeout 600 python bench.py --config $c --steps 150 > $R/bench_c$c.json 2> $R/bench_c$c.err; done
timeout 600 python bench.py --config 2 --envs 2048 --steps 158 --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --aux-rccl 0 > $R/bench_c2_2048envs.json 2> $R/bench_c2_2048envs.err
fi
if [ -z "$SKIP_SOAK" ]; then
timeout 600 python tools/gpu/random_soak.py 250 > $R/soak_random.log 2>&1; tail -1 $R/soak_random.log
timeout 900 python tools/gpu/soak.py 0.5 hull > $R/soak_noise05_hull.log 2>&1; tail -1 $R/soak_noise05_hull.log
fi
cd /tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --aux-fp32 0 --host-io 0 --aux-fingertips 0 --aux-large-hulls 0 --aux-rccl 0"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats -- $BENCH --steps 158 --warmup 20 > $R/stats.log 2>&1
SHORT="$BENCH --stagger 0 --steps 4 --warmup 1"
export RP_STREAM_SLICES=1
export RP_SPLIT_POS=0   # (the PMC passes: one slice, one-kernel position stage, as in rounds 3-5: comparable per-launch figures)
timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/fetch -- $SHORT > $R/fetch.log 2>&1
timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/write -- $SHORT > $R/write.log 2>&1
timeout 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/sq1 -- $SHORT > $R/sq1.log 2>&1
timeout 500 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d $R/sq2 -- $SHORT > $R/sq2.log 2>&1
timeout 500 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_BRANCH --output-format csv -d $R/sq3 -- $SHORT > $R/sq3.log 2>&1
export RP_SPLIT_POS=1
timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/fetch_split -- $SHORT > $R/fetch_split.log 2>&1
timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/write_split -- $SHORT > $R/write_split.log 2>&1
unset RP_STREAM_SLICES RP_SPLIT_POS
cd $GRAFT_REPO_ROOT
python tools/gpu/timeline.py $R/stats 260 > $R/timeline.txt 2>&1
python tools/gpu/summarize_profiles.py $R 2>&1 | tail -40
mkdir -p $R/summary; cp $R/timeline.txt $R/summary/r06_timeline.txt
for d in stats fetch write sq1 sq2 sq3 fetch_split write_split; do rm -rf $R/$d; done
du -sh $R
