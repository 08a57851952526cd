"""Per-kernel totals of one rocprofv3 --pmc pass (FETCH_SIZE / WRITE_SIZE in KB per dispatch; FETCH_SIZE doubled per
MI355X_MICROARCH.md: gfx950 tallies 128-B requests at 64 B):  python tools/pmc_sum.py pmc_counter_collection.csv COUNTER [top N]"""
import collections
import csv
import sys

path, counter = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
scale = 2048.0 if counter == 'FETCH_SIZE' else 1024.0
tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
for r in csv.DictReader(open(path)):
    if r['Counter_Name'] == counter:
        tot[r['Kernel_Name']] += float(r['Counter_Value']) * scale
        cnt[r['Kernel_Name']] += 1
print('%s, bytes (%s): per-kernel dispatches, MB per dispatch, MB total' % (counter, 'x2 corrected' if counter == 'FETCH_SIZE' else 'as counted'))
for k in sorted(tot, key=tot.get, reverse=True)[:top]:
    print('%6d  %10.2f  %10.1f  %s' % (cnt[k], tot[k] / cnt[k] / 1e6, tot[k] / 1e6, k[:110]))
print('all kernels: %.1f MB over %d dispatches' % (sum(tot.values()) / 1e6, sum(cnt.values())))
