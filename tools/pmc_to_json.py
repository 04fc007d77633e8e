"""rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate runs of `bench.py --workload W`) -> profiles/pmc_traffic.json, the file
bench.py reads `roofline.traffic` from.  HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE / WRITE_SIZE are in KB and on
gfx950 FETCH_SIZE counts half the bytes of a wide coalesced read (/opt/skills/guides/MI355X_MICROARCH.md, HBM section).

    python tools/pmc_to_json.py <dir with the counter_collection CSVs> <workload> <out.json> [note]"""
import collections
import csv
import glob
import json
import sys

src, workload, dst = sys.argv[1:4]
note = sys.argv[4] if len(sys.argv) > 4 else ''
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(src + '/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0].replace('void ', '').split('<')[0].strip()
        if not name.startswith('k_'):
            continue
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
kernels = {}
for name, d in sorted(acc.items()):
    if 'FETCH_SIZE' not in d or 'WRITE_SIZE' not in d:
        continue
    fk, wk = sum(d['FETCH_SIZE']) / len(d['FETCH_SIZE']), sum(d['WRITE_SIZE']) / len(d['WRITE_SIZE'])
    kernels[name] = {'launches_sampled': len(d['FETCH_SIZE']), 'fetch_size_kb_per_launch': fk, 'write_size_kb_per_launch': wk,
                     'hbm_bytes_per_launch': (2 * fk + wk) * 1024}
json.dump({'workload': workload, 'note': note, 'formula': 'hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, mean over the launches of the kernel (all template instances)',
           'kernels': kernels}, open(dst, 'w'), indent=1)
print('wrote', dst, len(kernels), 'kernels')
