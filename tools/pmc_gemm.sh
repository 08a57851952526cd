#!/bin/bash
# Hardware-counter passes over the bench step (one rocprofv3 run per counter set; counters only, with
# --kernel-trace for the kernel names).  Output: gpurun_out/pmc_<set>/..._counter_collection.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$i -o pmc -- \
    python3 $R/bench.py --no-cpu-baseline --no-profile --no-dense --no-pcie --no-configs --no-parity-check --steps 3 --warmup 2 > $R/gpurun_out/pmc_$i.log 2>&1
  echo "set $i ($set): rc=$?"
done
