#!/bin/bash
# quick GPU check: parity tests, phase profile, bench numbers
timeout 500 python -m pytest tests -x -q -m gpu -s 2>&1 | grep -E "replay rel|passed|failed|Error|error" | cut -c1-200 | tail -8
python scratch/phase_prof.py 64 4096 2>&1 | tail -27 | head -${1:-14}
python bench.py --no-cpu-baseline --aux-fp32 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp64 env-steps/s %.0f step-seq ms %.3f solver launch ms %.3f | fp32 %.0f step-seq ms %.3f' % (d['value'], d['roofline']['step_sequence_avg_ms'], d['roofline']['kernel_avg_ms'], d['aux']['fp32_engine']['value'], d['aux']['fp32_engine']['kernel_avg_ms']))"
