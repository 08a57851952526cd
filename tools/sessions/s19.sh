#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s19
mkdir -p $O
cd $R
timeout 600 tools/micro/p2_bench.bin 7221 20 > $O/p2_bench.txt 2>&1; echo "rc=$?"
grep -v "^  [ 0-9]" $O/p2_bench.txt | head -90
