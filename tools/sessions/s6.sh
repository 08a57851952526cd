#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s6
mkdir -p $O
cd $R
for g in "" "--main-priority -1" "" "--main-priority -1"; do
timeout 600 python3 bench.py --no-cpu-baseline --no-pcie --no-configs --no-strict --no-dense --no-eval --no-profile $g > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('[$g]', d['value'], d['ms_per_step'])
PY
done
