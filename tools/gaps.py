"""Per-step gaps between consecutive kernels from a rocprofv3 kernel trace CSV."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'elastic_convpool_fwd' in r['Kernel_Name']]
for st in range(-6, -2):
    seg = rows[idx[st]:idx[st + 1] + 1]
    print(" ".join("%s:%.1f|%.1f" % (a['Kernel_Name'].replace('void ', '')[:10], (int(a['End_Timestamp']) - int(a['Start_Timestamp'])) / 1e3,
                                     (int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3) for a, b in zip(seg[:-1], seg[1:])))
