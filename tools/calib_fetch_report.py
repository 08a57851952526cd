import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.OrderedDict()
for r in rows:
    k = (r['Kernel_Name'][:70], r['Grid_Size'], r['Counter_Name'])
    acc.setdefault(k, []).append(float(r['Counter_Value']))
for k, v in acc.items():
    print('%-72s grid %-9s %-28s n=%d  mean %.1f' % (k[0], k[1], k[2], len(v), sum(v) / len(v)))
