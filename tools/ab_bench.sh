#!/bin/bash
# A/B of bench.py flag sets in one GPU visit: bash tools/ab_bench.sh "<flags A>" "<flags B>" ...   (prints value / ms per step)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for f in "$@"; do
  python3 $R/bench.py --no-cpu-baseline --no-pcie --no-configs --no-dense --no-parity-check $f 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-40s %9.1f clips/s  %.3f ms  %s  host %.3f ms' % (sys.argv[1], d['value'], d['ms_per_step'], d['config']['step_launch'], d['host_enqueue_ms_per_step']))" "$f"
done
done
