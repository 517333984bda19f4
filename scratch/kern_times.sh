#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --no-cpu-baseline --aux-fp32 0 --precision ${1:-32} --steps 40 --warmup 10 > /dev/null 2>&1
python - <<'PY'
import pandas as pd, glob
ks=pd.read_csv(glob.glob('/tmp/kt/*/*kernel_stats.csv')[0])
for _,r in ks[ks.Name.str.contains('rp_stage')].iterrows(): print(r.Name[:45], r.Calls, '%.1f us' % (r.AverageNs/1e3))
PY
