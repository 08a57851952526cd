#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s13
mkdir -p $O
cd $R
for i in 1 2; do
timeout 600 python3 bench.py --no-cpu-baseline --no-configs --no-strict --no-eval --no-pcie --no-dense > $O/bench.json 2> $O/bench.err
python3 - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
k=d['kernels']
print(d['value'], d['ms_per_step'], 'stage', k['stage']['avg_ms'], k['stage'].get('alone_avg_ms'), 'K1', k['embed_l1_fwd']['avg_ms'])
PY
done
