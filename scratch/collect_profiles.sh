#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats of the bench command, then separate PMC passes.
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf $R/gpurun_out/prof_r01; mkdir -p $R/gpurun_out/prof_r01
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01/stats -- python $R/bench.py --no-cpu-baseline --aux-fp32 0 --host-io 0 > $R/gpurun_out/prof_r01/bench_under_rocprof.json 2> $R/gpurun_out/prof_r01/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_r01/fetch -- python $R/bench.py --no-cpu-baseline --aux-fp32 0 --host-io 0 --steps 20 --warmup 5 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_r01/write -- python $R/bench.py --no-cpu-baseline --aux-fp32 0 --host-io 0 --steps 20 --warmup 5 > /dev/null 2>&1
cd $R && python bench.py > gpurun_out/prof_r01/bench_plain.json 2> gpurun_out/prof_r01/bench_plain.err
ls -R gpurun_out/prof_r01 | head -30
