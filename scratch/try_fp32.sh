#!/bin/bash
# fp32 bench with alternative libs (the stage hand-over layout depends on RPK_NE: whole lib swapped)
cp robopianist_amd/csrc/librp_engine.so /tmp/keep.so
for f in robopianist_amd/csrc/librp_engine.so "$@"; do
  [ "$f" != robopianist_amd/csrc/librp_engine.so ] && cp $f robopianist_amd/csrc/librp_engine.so
  echo "== $f"
  RP_SKIP_SELF_CHECK=1 python scratch/sanity.py 2>&1 | tail -1 | cut -c1-120
  RP_SKIP_SELF_CHECK=1 python bench.py --precision 32 --no-cpu-baseline --aux-fp32 0 --host-io 0 --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   fp32 env-steps/s %.0f  solver %.3f ms  seq %.3f ms warn %s' % (d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['step_sequence_avg_ms'], d['sanity']))"
done
cp /tmp/keep.so robopianist_amd/csrc/librp_engine.so
