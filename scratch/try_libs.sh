#!/bin/bash
# try alternative builds of librp_engine.so: sanity vs oracle, then bench
cp robopianist_amd/csrc/librp_engine.so /tmp/keep.so
for f in "$@"; do
  cp $f robopianist_amd/csrc/librp_engine.so
  echo "== $f"
  RP_SKIP_SELF_CHECK=1 python scratch/sanity.py 2>&1 | tail -2 | cut -c1-100
  python bench.py --no-cpu-baseline --aux-fp32 0 --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   env-steps/s %.0f  solver %.3f ms' % (d['value'], d['roofline']['kernel_avg_ms']))"
done
cp /tmp/keep.so robopianist_amd/csrc/librp_engine.so
