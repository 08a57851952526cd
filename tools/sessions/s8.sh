#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s8
mkdir -p $O
cd $R
timeout 900 python3 bench.py --no-cpu-baseline --no-configs --no-strict --no-eval > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.err
python3 - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
print('q32', d['q32_storage'])
fa=d['feature_assembly']
for k,v in fa.items():
    if isinstance(v,dict) and 'value' in v: print(k, v['value'], v['ms_per_step'])
print(fa['training_entry_point'])
print(fa['first_step_parity'])
PY
python3 tools/pieces_step.py resident 30 --sites 2>/dev/null | tail -25
