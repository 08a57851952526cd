#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s28
mkdir -p $O
cd $R
for cap in 4096 1024 2048 512 8192 1024 4096; do
LIREC_STAGE_BLOCKS=$cap timeout 600 python3 bench.py --no-cpu-baseline --no-configs --no-strict --no-eval --no-pcie --no-dense > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 200 $O/bench.err
python3 - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
K=d['kernels']
print('[cap $cap]', d['value'], d['ms_per_step'], 'stage', K['stage']['avg_ms'], K['stage'].get('alone_avg_ms'), 'l1', K['embed_l1_fwd']['avg_ms'])
PY
done
