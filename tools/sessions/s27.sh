#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s27
mkdir -p $O
cd $R
for g in "" "--pipeline 1" "" "--pipeline 1" "" "--pipeline 1"; do
timeout 600 python3 bench.py --no-cpu-baseline --no-configs --no-strict --no-eval --no-pcie --no-dense $g > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 200 $O/bench.err
python3 - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
K=d['kernels']
print('[$g]', d['value'], d['ms_per_step'], 'roof', d['roofline']['site'], d['roofline']['frac'], 'l1', K['embed_l1_fwd']['avg_ms'], 'stage', K['stage']['avg_ms'], 'dW1', K['embed_dW1']['avg_ms'], 'host', d.get('host_enqueue_ms_per_step'))
PY
done
