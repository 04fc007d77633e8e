"""Median / min duration per (kernel, grid) from a rocprofv3 --kernel-trace --output-format csv directory.  python tools/ktrace_medians.py <dir> [name filter]"""
import collections
import csv
import glob
import sys
fs = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
flt = sys.argv[2] if len(sys.argv) > 2 else ''
d = collections.defaultdict(list)
order = []
for f in fs:
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if flt in n:
            k = (n[:56], r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', ''))
            if k not in d:
                order.append(k)
            d[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000.0)
for k in order:
    v = sorted(d[k])
    print('%-58s grid %8s wg %5s  n %4d  median %8.1f us  min %8.1f' % (k[0], k[1], k[2], len(v), v[len(v) // 2], v[0]))
