#!/bin/bash
# Same-box A/B of two builds of the library: bench.py with LIREC_LIB_PATH=<lib A> / <lib B> in turn, N alternating rounds.
#   bash tools/ab_lib.sh lirec_amd/liblirec_hip_base.so lirec_amd/liblirec_hip.so [rounds] [extra bench flags]
R=${GRAFT_REPO_ROOT:-$(pwd)}; A=$1; B=$2; N=${3:-3}; shift 3
Q="--no-cpu-baseline --no-configs --no-pcie --no-strict --no-eval --no-dense --no-parity-check"
for r in $(seq 1 $N); do
  for L in $A $B; do
    LIREC_LIB_PATH=$R/$L timeout 600 python3 $R/bench.py $Q "$@" 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read()); k = d['kernels']
print('[%s] %9.1f clips/s  %.3f ms' % (sys.argv[1], d['value'], d['ms_per_step']))
print('    ' + '  '.join('%s %.0f/%.0f' % (n, 1e3 * v['avg_ms'], 1e3 * (v.get('alone_avg_ms') or 0)) for n, v in k.items()))" $L
  done
done
