#!/bin/bash
# GPU session: gate q32 path
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s2
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -k "gate_on_staged" > $O/pytest_gate.log 2>&1; echo "pytest gate rc=$?"; tail -15 $O/pytest_gate.log
timeout 1500 python3 -m pytest tests/test_gpu_recorded_bench_shape.py tests/test_gpu_bench_shape.py tests/test_gpu_planes.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
for g in 1 0; do
timeout 600 python3 bench.py --no-cpu-baseline --no-pcie --no-configs --no-strict --no-dense --no-eval --set gate_q32=$g > $O/bench_g$g.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.err
python3 - <<PY
import json
d=json.loads(open('$O/bench_g$g.json').read().strip().splitlines()[-1])
print('gate_q32=$g', d['value'], d['ms_per_step'])
for k,v in d['kernels'].items(): print('  ', k, v['avg_ms'], v.get('alone_avg_ms'), v['frac'])
PY
done
