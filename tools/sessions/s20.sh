#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s20
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_planes.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
timeout 900 python3 bench.py --no-cpu-baseline --no-strict --no-pcie --no-dense > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.err
python3 - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['site'], d['roofline']['frac'])
print('eval', json.dumps(d['eval'])[:900])
for c in d['configs'] or []:
    print(c.get('config'), c.get('value'), c.get('ms_per_step'), c.get('layer1'), (c.get('roofline') or {}).get('site'), (c.get('roofline') or {}).get('frac'), (c.get('roofline') or {}).get('avg_launch_ms'), c.get('error'))
PY
