#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s25
mkdir -p $O
cd $R
for g in "" "--set gate_stage_after_step_start=1" "" "--set gate_stage_after_step_start=1"; do
timeout 600 python3 bench.py --no-cpu-baseline --no-configs --no-strict --no-eval --no-pcie --no-dense $g > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 200 $O/bench.err
python3 - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
K=d['kernels']
print('[$g]', d['value'], d['ms_per_step'], 'l1', K['embed_l1_fwd']['avg_ms'], 'stage', K['stage']['avg_ms'], 'dW1', K['embed_dW1']['avg_ms'], 'gate_stage', K['gate_stage']['avg_ms'], 'adam', K['adam']['avg_ms'])
PY
done
