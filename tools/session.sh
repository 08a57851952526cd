#!/bin/bash
# One GPU-box visit (through gpurun), parameterised -- what the 29 per-experiment scripts of round 4 (tools/sessions/s*.sh, now
# removed; HISTORY.md keeps what they measured) each spelled out by hand:
#   bash tools/session.sh <tag> [--tests "<pytest args>"] [--micro <name> ...] [--trace] [--reps N] [--ab "<bench flags A>" "<bench flags B>" ...]
#   --tests : a pytest selection, e.g. "tests/test_gpu_ops.py -k gate"          -> gpurun_out/<tag>/pytest.log
#   --micro : tools/micro/<name>.hip built here and run                        -> gpurun_out/<tag>/<name>.txt
#   --trace : one step's kernel timeline (tools/trace_step.sh)                 -> gpurun_out/<tag>/trace/trace.csv
#   --ab    : bench.py with each flag set in turn, N times round-robin (same box, alternating: box-to-box spread is ~2 %),
#             clips/s, ms per step and the per-site times in the step / alone   -> gpurun_out/<tag>/ab.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-session}; shift
O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
REPS=2; TESTS=""; MICRO=(); TRACE=0; AB=()
while [ $# -gt 0 ]; do
  case "$1" in
    --tests) TESTS="$2"; shift 2;;
    --micro) MICRO+=("$2"); shift 2;;
    --trace) TRACE=1; shift;;
    --reps) REPS=$2; shift 2;;
    --ab) shift; while [ $# -gt 0 ]; do AB+=("$1"); shift; done;;
    *) echo "unknown argument $1"; exit 2;;
  esac
done
if [ -n "$TESTS" ]; then
  timeout 2400 python3 -m pytest $TESTS -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
fi
for m in "${MICRO[@]}"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I lirec_amd/csrc -I include tools/micro/$m.hip -o /tmp/$m.bin 2> $O/$m.build.log \
    && timeout 600 /tmp/$m.bin > $O/$m.txt 2>&1; echo "$m rc=$?"; cat $O/$m.txt
done
[ $TRACE -eq 1 ] && bash tools/trace_step.sh $TAG/trace > /dev/null 2>&1 && cat $O/trace/trace.csv
Q="--no-cpu-baseline --no-configs --no-pcie --no-strict --no-eval --no-dense --no-parity-check"
: > $O/ab.txt
for rep in $(seq 1 $REPS); do
  for f in "${AB[@]}"; do
    timeout 600 python3 bench.py $Q $f 2> $O/bench.err | python3 -c "
import json, sys
d = json.loads(sys.stdin.read()); k = d['kernels']
print('[%s] %9.1f clips/s  %.3f ms' % (sys.argv[1], d['value'], d['ms_per_step']))
print('    ' + '  '.join('%s %.0f/%.0f' % (n, 1e3 * v['avg_ms'], 1e3 * (v.get('alone_avg_ms') or 0)) for n, v in k.items()))" "$f" | tee -a $O/ab.txt
  done
done
