"""Per-dispatch durations of one kernel from a rocprofv3 kernel trace: python tools/trace_kernel.py <dir> <substr>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if sys.argv[2] in r['Kernel_Name']]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
print(len(d), 'dispatches; last 12 (us):', [round(x, 1) for x in d[-12:]])
