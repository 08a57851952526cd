#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s24
mkdir -p $O
cd $R
t0=$(date +%s)
timeout 1200 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? wall $(( $(date +%s) - t0 )) s"; tail -c 300 $O/bench.err
python3 - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['site'], d['roofline']['frac'], d['roofline'].get('mfma_pipe_busy'), d['roofline'].get('traffic_source'))
for c in d['configs'] or []:
    print(c.get('config')[:50], c.get('value'), c.get('ms_per_step'), c.get('step_launch')[:20], (c.get('roofline') or {}).get('site'), (c.get('roofline') or {}).get('frac'))
PY
