#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s4
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_recorded_bench_shape.py tests/test_gpu_loops.py tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for g in "gate_stage_on_side=1" "gate_stage_on_side=0" "gate_q32=0"; do
timeout 600 python3 bench.py --no-cpu-baseline --no-pcie --no-configs --no-strict --no-dense --no-eval --set $g > $O/bench_$g.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.err
python3 - <<PY
import json
d=json.loads(open('$O/bench_$g.json').read().strip().splitlines()[-1])
print('$g', d['value'], d['ms_per_step'])
for k,v in d['kernels'].items():
    if k.startswith('gate') or k in ('pool_bwd','embed_dZ1','adam','stage','embed_l1_fwd'): print('  ', k, v['avg_ms'], v.get('alone_avg_ms'), v['frac'])
PY
done
