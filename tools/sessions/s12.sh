#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s12
mkdir -p $O
cd $R
run() {
  timeout 600 python3 bench.py --no-cpu-baseline --no-configs --no-strict --no-eval --no-pcie --no-dense --no-profile "$@" > $O/bench.json 2> $O/bench.err; 
  python3 - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('$LIREC_PRESTAGE_AT [$*]', d['value'], d['ms_per_step'], d['config']['input_pipeline'][:12])
PY
}
export LIREC_PRESTAGE_AT=loss; run
export LIREC_PRESTAGE_AT=start; run
export LIREC_PRESTAGE_AT=start; run --main-priority -1 --set side_stream_priority=-1
export LIREC_PRESTAGE_AT=loss; run --main-priority -1 --set side_stream_priority=-1
run --pipeline 0
run --pipeline 0 --main-priority -1 --set side_stream_priority=-1
