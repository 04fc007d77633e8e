"""rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate runs of `bench.py --workload W --dtype D --width N`) -> one entry of
profiles/pmc_traffic.json, the file bench.py reads `roofline.traffic`, `roofline.kernels.differentiable.traffic` and `roofline.step.hbm_bytes` from.
HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE / WRITE_SIZE are in KB and on gfx950 FETCH_SIZE counts half the bytes of a wide
coalesced read (/opt/skills/guides/MI355X_MICROARCH.md, HBM section).

    python tools/pmc_to_json.py <dir with the counter_collection CSVs> <workload> <dtype> <width> <out.json> [note] [scaling] [rays per GPU]

The entry key is "<workload>|<dtype>|<width>|<scaling>" (scaling: weak | strong, default weak) and the entry records the rays per GPU of the profiled command:
bench.py returns no traffic for a line whose command differs in either.  Other entries of an existing <out.json> are kept."""
import collections
import csv
import glob
import json
import os
import sys

src, workload, dtype, width, dst = sys.argv[1:6]
note = sys.argv[6] if len(sys.argv) > 6 else ''
scaling = sys.argv[7] if len(sys.argv) > 7 else 'weak'
rays = int(sys.argv[8]) if len(sys.argv) > 8 else None
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(src + '/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0].replace('void ', '').split('<')[0].strip()
        if not name.startswith('k_'):
            continue
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
steps = len(acc['k_step_prologue']['FETCH_SIZE']) if 'k_step_prologue' in acc else 0      # launched once per step
assert steps > 0, 'no k_step_prologue launches in ' + src
kernels = {}
for name, d in sorted(acc.items()):
    if 'FETCH_SIZE' not in d or 'WRITE_SIZE' not in d:
        continue
    fk, wk = sum(d['FETCH_SIZE']) / len(d['FETCH_SIZE']), sum(d['WRITE_SIZE']) / len(d['WRITE_SIZE'])
    kernels[name] = {'launches_sampled': len(d['FETCH_SIZE']), 'launches_per_step': len(d['FETCH_SIZE']) / steps, 'fetch_size_kb_per_launch': fk,
                     'write_size_kb_per_launch': wk, 'hbm_bytes_per_launch': (2 * fk + wk) * 1024}
doc = {}
if os.path.exists(dst):
    try:
        doc = json.load(open(dst))
    except ValueError:
        doc = {}
if 'entries' not in doc:
    doc = {'formula': 'hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, mean over the launches of the kernel (all template instances); '
                      'launches_per_step = launches sampled / launches of k_step_prologue (one per step)', 'entries': {}}
doc['entries']['%s|%s|%s|%s' % (workload, dtype, width, scaling)] = {'note': note, 'steps_sampled': steps, 'rays_per_gpu': rays, 'kernels': kernels}
json.dump(doc, open(dst, 'w'), indent=1)
print('wrote', dst, '%s|%s|%s|%s' % (workload, dtype, width, scaling), len(kernels), 'kernels,', steps, 'steps')
