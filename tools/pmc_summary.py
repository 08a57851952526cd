"""Average the counters of gpurun_out/pmc_*/pmc_counter_collection.csv per kernel (diagnostics)."""
import csv, collections, glob, sys
pat = sys.argv[1:] or ['gemm_bf16x3_kernel<0, 3, 1', 'gemm_bf16x3_kernel<2, 2, 3']
res = collections.defaultdict(dict)
for f in sorted(glob.glob('gpurun_out/pmc_*/pmc_counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:64]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in agg.items():
        for c, v in d.items():
            res[k][c] = sum(v) / len(v)
for k, d in res.items():
    if any(p in k for p in pat):
        print(k)
        for c, v in sorted(d.items()):
            print('   %-32s %16.0f' % (c, v))
