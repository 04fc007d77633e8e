"""Per-kernel means of the counters in a rocprofv3 --pmc output directory (dev tool): python tools/pmc_ro_summary.py <dir> [name filter]"""
import csv, glob, sys, collections
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else 'k_sdf_col0'
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if flt in r['Kernel_Name']:
            acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(k)
    for c, vals in sorted(v.items()):
        print('   %-28s %.4g (n=%d)' % (c, sum(vals) / len(vals), len(vals)))
