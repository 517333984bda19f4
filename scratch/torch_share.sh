#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf $R/gpurun_out/ts; mkdir -p $R/gpurun_out/ts
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ts/e -- python $R/bench.py --no-cpu-baseline --aux-fp32 0 --graph 0 --steps 60 --warmup 10 > $R/gpurun_out/ts/eager.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ts/g -- python $R/bench.py --no-cpu-baseline --aux-fp32 0 --graph 1 --steps 60 --warmup 10 > $R/gpurun_out/ts/graph.json 2>/dev/null
cd $R
python bench.py --no-cpu-baseline --aux-fp32 0 --graph 0 | tail -1 | cut -c1-200
python bench.py --no-cpu-baseline --aux-fp32 0 --graph 1 | tail -1 | cut -c1-200
python - <<'PY'
import pandas as pd, glob
for d in ('e','g'):
    ks=pd.read_csv(glob.glob(f'gpurun_out/ts/{d}/*/*kernel_stats.csv')[0])
    n=70 if d=='e' else 70
    t=ks[~ks.Name.str.contains('rp_stage')]
    print(d, 'torch launches/step %.0f  torch us/step %.0f  engine us/step %.0f' % (t.Calls.sum()/n, t.TotalDurationNs.sum()/n/1e3, ks[ks.Name.str.contains('rp_stage')].TotalDurationNs.sum()/n/1e3))
PY
