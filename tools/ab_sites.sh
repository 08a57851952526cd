#!/bin/bash
# per-site launch times of the bench step under several flag sets (one GPU visit)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for f in "$@"; do
  python3 $R/bench.py --no-cpu-baseline --no-pcie --no-configs --no-dense --no-parity-check --steps 40 $f 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels']
print('%-28s %.3f ms | ' % (sys.argv[1], d['ms_per_step']) + ' '.join('%s %.0f' % (n.replace('embed_','e_').replace('linear_','l_'), v['avg_ms']*v['launches_per_step']*1000) for n,v in k.items()))" "$f"
done
