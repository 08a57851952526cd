#!/bin/bash
# PMC passes over the layer-1 micro-benchmark (full kernels only).  Usage: tools/micro/p2_pmc.sh [ablation index, default 0]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
AB=${1:-0}
OUT=$R/gpurun_out/p2_pmc_$AB
rm -rf $OUT; mkdir -p $OUT
pass() { n=$1; shift; rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/$n -o p -- $R/tools/micro/p2_bench.bin 7096 5 $AB > $OUT/$n.log 2>&1; }
pass a SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
pass b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
pass c GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
pass d FETCH_SIZE
python3 - <<PY
import csv, glob, collections
for n in 'abcd':
    for f in glob.glob('$OUT/%s/**/*counter_collection.csv' % n, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:60]
            if 'gemm_p2' not in k: continue
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
        for k in acc:
            print(n, k)
            for c, v in sorted(acc[k].items()): print('    %-28s %.5g' % (c, v / max(cnt[(k, c)], 1)))
PY
