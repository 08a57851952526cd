#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s10
mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.err
python3 - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['site'], d['roofline']['frac'])
print('q32', d['q32_storage'])
print('strict', d['strict_f32']['value'], 'dense', d['dense_fill']['value'], 'eval', d['eval']['value'])
fa=d['feature_assembly']
for k,v in fa.items():
    if isinstance(v,dict) and 'value' in v: print(k, v['value'], v['ms_per_step'])
print(fa['training_entry_point'])
for c in d['configs']: print(c['config'][:40], c['value'], c['ms_per_step'], c['roofline']['site'], c['roofline']['frac'])
print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
