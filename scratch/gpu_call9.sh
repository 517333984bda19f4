#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/prof_r02b
mkdir -p $R
cd $GRAFT_REPO_ROOT
( python -m pytest tests -m gpu -x -q ) > $R/pytest_gpu.log 2>&1
tail -6 $R/pytest_gpu.log
python bench.py > $R/bench_plain.json 2> $R/bench_plain.err
for c in 3 4 5; do python bench.py --config $c --steps 150 --no-cpu-baseline > $R/bench_c$c.json 2> $R/bench_c$c.err; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --aux-fp32 0 --host-io 0 > $R/bench_driver_window.json 2>&1
cd /tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --aux-fp32 0 --host-io 0 --stagger 0 --steps 4 --warmup 1"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/write -- $BENCH > $R/write.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/fetch -- $BENCH > $R/fetch.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/sq1 -- $BENCH > $R/sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d $R/sq2 -- $BENCH > $R/sq2.log 2>&1
cd $GRAFT_REPO_ROOT
du -sh $R
for f in $R/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    print(sys.argv[1].split('/')[-1], "value %.0f"%j["value"], "solver_ms %.4f"%j["roofline"]["kernel_avg_ms"], "seq %.3f"%j["roofline"]["step_sequence_avg_ms"], {k:(round(v.get('value')) if isinstance(v,dict) and v.get('value') else None) for k,v in j.get("aux",{}).items()}, j["sanity"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
