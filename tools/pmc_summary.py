"""Mean of every PMC counter per kernel from rocprofv3 --pmc counter_collection CSVs under a directory."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0][:48]
        if len(sys.argv) > 2 and sys.argv[2] not in name: continue
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()): print(f'   {c:34s} mean {sum(v)/len(v):16.1f}  n={len(v)}')
