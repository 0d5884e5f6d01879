"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel dispatch."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r['Kernel_Name'].split('(')[0][-60:]
    acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for k, c in acc.items():
    if flt and flt not in k:
        continue
    print(k)
    for cn, v in sorted(c.items()):
        print('   %-28s n=%-4d mean=%.4g' % (cn, len(v), sum(v) / len(v)))
