"""Average duration of every tracer launch of a step from a rocprofv3 --kernel-trace CSV (launches of one kernel grouped by their
position inside the step)."""
import csv, glob, sys, collections
path = sys.argv[1]
f = glob.glob(path + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'].split('(')[0][:60], int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in rows]
# split into steps at k_sphere_trace
steps, cur = [], None
for name, d in seq:
    if 'k_sphere_trace' in name:
        if cur: steps.append(cur)
        cur = []
    if cur is not None: cur.append((name, d))
steps = steps[len(steps) // 4:]                                   # drop warmup
acc = collections.OrderedDict()
for st in steps:
    seen = collections.Counter()
    for name, d in st:
        if not any(k in name for k in ('k_ray_samples', 'k_sphere_trace', 'k_reduce_items')): continue
        key = '%s #%d' % (name, seen[name]); seen[name] += 1
        acc.setdefault(key, []).append(d)
for k, v in acc.items():
    print(f'{k:70s} {sum(v) / len(v) / 1e3:9.1f} us  (n={len(v)})')
