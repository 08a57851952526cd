#!/bin/bash
# GPU session 1 of round 4
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s1
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_recorded_bench_shape.py tests/test_gpu_planes.py tests/test_gpu_ops.py tests/test_gpu_loops.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
timeout 300 tools/micro/l2_lds_intake.bin > $O/intake.txt 2>&1; echo "intake rc=$?"; cat $O/intake.txt
timeout 600 python3 bench.py --no-cpu-baseline --no-pcie --no-configs --no-strict > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 400 $O/bench.err
python3 - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'])
for k,v in d['kernels'].items(): print(k, v['avg_ms'], v.get('alone_avg_ms'), v['frac'])
PY
