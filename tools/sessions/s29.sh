#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s29
mkdir -p $O
cd $R
timeout 600 python3 tools/diag_unpool.py > $O/diag.log 2>&1; echo "diag rc=$?"; tail -6 $O/diag.log
for g in "" "--ablate 32768" "" "--ablate 32768"; do
timeout 600 python3 bench.py --no-cpu-baseline --no-configs --no-strict --no-eval --no-pcie --no-dense $g > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 200 $O/bench.err
python3 - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
K=d['kernels']
print('[$g]', d['value'], d['ms_per_step'], 'pool_bwd', K['pool_bwd']['avg_ms'], K['pool_bwd'].get('alone_avg_ms'), K['pool_bwd']['frac'], 'dW1', K['embed_dW1']['avg_ms'])
PY
done
