import csv, sys, glob
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print('%-70s calls=%-4s avg_us=%8.2f min=%8.2f max=%8.2f pct=%s' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, r['Percentage']))
