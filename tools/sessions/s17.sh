#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s17
mkdir -p $O
cd $R
i=0
for g in "" "--set fuse_dw1_adam=0" "--pipeline 1" "--pipeline 1 --set fuse_dw1_adam=0" "" "--set fuse_dw1_adam=0"; do
i=$((i+1))
timeout 600 python3 bench.py --no-cpu-baseline --no-configs --no-strict --no-eval --no-pcie --no-profile $g > $O/bench$i.json 2> $O/bench$i.err; echo "bench rc=$?"; tail -c 200 $O/bench$i.err
python3 - <<PY
import json
d=json.loads(open('$O/bench$i.json').read().strip().splitlines()[-1])
print('[$g]', d['value'], d['ms_per_step'], 'q32', (d.get('q32_storage') or {}).get('value'), (d.get('q32_storage') or {}).get('site_ms'), 'plain', (d.get('no_input_pipeline') or {}).get('value'))
PY
done
