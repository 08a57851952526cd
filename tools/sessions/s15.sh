#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s15
mkdir -p $O
cd $R
timeout 2000 python3 -m pytest tests/test_gpu_recorded_bench_shape.py tests/test_gpu_planes.py tests/test_gpu_loops.py tests/test_pieces_entry.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
for g in ""; do
timeout 600 python3 bench.py --no-cpu-baseline --no-configs --no-strict --no-eval --no-pcie --no-dense $g > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.err
python3 - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('[$g]', d['value'], d['ms_per_step'], d['roofline']['site'], d['roofline']['frac'])
for k,v in d['kernels'].items(): print('  ', k, v['avg_ms'], v.get('alone_avg_ms'), v['frac'])
PY
done
