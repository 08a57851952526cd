#!/bin/bash
# One GPU-box visit: the gpu test suite, the default bench line, a kernel trace of the bench command.
# usage (through gpurun): bash tools/gpu_round.sh <tag> [pytest-args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-x}; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests -m gpu -x -q "$@" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
cp gpurun_out/parity_errors.json $O/ 2>/dev/null
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --no-cpu-baseline --no-profile --no-dense --no-pcie --no-parity-check --no-configs --steps 50 --warmup 10 > $O/kt.log 2>&1; echo "kernel-trace rc=$?"
cp $O/kt/kt_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
rm -rf $O/kt/*_kernel_trace.csv 2>/dev/null
head -c 1500 $O/bench.json
