#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s18
mkdir -p $O
cd $R
LIREC_PRESTAGE_AT=gate timeout 900 python3 -m pytest tests/test_gpu_recorded_bench_shape.py -x -q -m gpu -k pipelined > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for at in gate start gate start loss; do
LIREC_PRESTAGE_AT=$at timeout 600 python3 bench.py --no-cpu-baseline --no-configs --no-strict --no-eval --no-pcie --no-dense --pipeline 1 > $O/bench_$at.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.err
python3 - <<PY
import json
d=json.loads(open('$O/bench_$at.json').read().strip().splitlines()[-1])
print('[$at]', d['value'], d['ms_per_step'], d['roofline']['site'], d['roofline']['frac'])
for k,v in d['kernels'].items():
    if k in ('stage','embed_l1_fwd','gate_fwd','gate_dEE','embed_dW1','loss','linear_fwd','linear_dA'): print('  ', k, v['avg_ms'], v.get('alone_avg_ms'), v['frac'])
PY
done
