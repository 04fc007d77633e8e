"""rocprofv3 --pmc counter_collection CSVs under a directory -> one summary CSV (kernel, counter, launches, mean, min, max).
Kernels launched several times per step with different work (k_ray_samples: 3 launches) are ALSO reported per position in the step."""
import csv, glob, sys, collections
src, dst = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in sorted(glob.glob(src + '/**/*counter_collection.csv', recursive=True)):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    pos = collections.Counter()
    seen_dispatch = {}
    for r in rows:
        name = r['Kernel_Name'].split('(')[0]
        if not (name.startswith('k_') or name.startswith('void k_')): continue
        acc[(name, r['Counter_Name'])].append(float(r['Counter_Value']))
        if 'k_ray_samples' in name:
            key = (r['Dispatch_Id'], name)
            if key not in seen_dispatch:
                seen_dispatch[key] = pos[name] % 3; pos[name] += 1
            acc[('%s [launch %d of the step]' % (name, seen_dispatch[key]), r['Counter_Name'])].append(float(r['Counter_Value']))
with open(dst, 'w', newline='') as fh:
    w = csv.writer(fh)
    w.writerow(['kernel', 'counter', 'launches', 'mean_per_launch', 'min', 'max'])
    for (k, c), v in sorted(acc.items()):
        w.writerow([k, c, len(v), sum(v) / len(v), min(v), max(v)])
print('wrote', dst, len(acc), 'rows')
