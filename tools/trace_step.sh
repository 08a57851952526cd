#!/bin/bash
# kernel timeline of a few bench steps (rocprofv3 --kernel-trace): gpurun_out/<tag>/trace.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-trace}; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --no-cpu-baseline --no-profile --no-dense --no-pcie --no-parity-check --no-configs --steps 6 --warmup 3 "$@" > $O/kt.log 2>&1
echo "rc=$?"
python3 - <<PY
import csv, glob
f = glob.glob('$O/kt/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# one full train step of the timed loop: from one step's first kernel (zero_grad + counters) to the next one's.  The step's OWN
# stream is the one layer 1 runs on; the weight-gradient side stream has a counter launch of its own (in front of its Adam
# launch, round 5), which is not a step boundary.
sid = lambda r: r.get('Stream_Id', r.get('Queue_Id', ''))
main = [sid(r) for r in rows if 'gemm_p2_nt_kernel' in r['Kernel_Name'] or 'stage_fused_kernel' in r['Kernel_Name']]
main = max(set(main), key=main.count) if main else sid(rows[-1])
idx = [i for i, r in enumerate(rows) if sid(r) == main and ('zero_count_kernel' in r['Kernel_Name'] or 'counter_add_kernel' in r['Kernel_Name'])]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]['Start_Timestamp'])
out = open('$O/trace.csv', 'w')
out.write('start_us,dur_us,gap_us,stream,kernel\n')
prev_end = t0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    out.write('%.1f,%.1f,%.1f,%s,%s\n' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get('Stream_Id', r.get('Queue_Id', '')), r['Kernel_Name'][:90].replace(',', ';')))
    prev_end = max(prev_end, e)
out.close()
print(open('$O/trace.csv').read())
PY
rm -rf $O/kt
