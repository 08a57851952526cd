#!/bin/bash
# FETCH_SIZE of the layer-1 forward launch under several bench.py flag sets: bash tools/ab_fetch.sh "<flags A>" "<flags B>"
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/abfetch; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for f in "$@"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p$i -o pmc -- python3 $R/bench.py --no-cpu-baseline --no-profile --no-dense --no-pcie --no-configs --no-parity-check --steps 5 --warmup 3 $f > $O/p$i.log 2>&1
  echo "== $f"
  python3 - $O/p$i/pmc_counter_collection.csv <<'P'
import csv, sys, collections
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'][:60]
    if 'gemm' in k:
        acc.setdefault(k, []).append(float(r['Counter_Value']))
for k, v in acc.items():
    print('  %-62s n=%-3d fetch %.1f MB (x2 rule)' % (k, len(v), 2 * sum(v) / len(v) / 1000.0))
P
  rm -rf $O/p$i
done
