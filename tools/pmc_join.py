"""Join rocprofv3 kernel_trace + counter_collection: per kernel mean duration and counters."""
import csv, sys, glob, collections
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ''
tr = glob.glob(d + '/*/*kernel_trace.csv'); cc = glob.glob(d + '/*/*counter_collection.csv')
dur = collections.defaultdict(list)
for r in csv.DictReader(open(tr[0])):
    dur[r['Kernel_Name'].split('(')[0][-50:]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc[0])):
    acc[r['Kernel_Name'].split('(')[0][-50:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k in acc:
    if flt and flt not in k: continue
    print('%s  n=%d dur_us=%.1f' % (k, len(dur[k]), sum(dur[k]) / max(1, len(dur[k]))))
    print('    ' + '  '.join('%s=%.4g' % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())))
