#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s11
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_recorded_bench_shape.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
timeout 900 python3 bench.py --no-cpu-baseline --no-configs --no-strict --no-eval --no-pcie > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench.err
python3 - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['input_pipeline'][:40], d['config']['step_launch'])
print('plain', d['no_input_pipeline'])
print('q32', d['q32_storage'])
for k,v in d['kernels'].items(): print('  ', k, v['avg_ms'], v.get('alone_avg_ms'), v['frac'])
PY
