"""gpurun_out/<tag> (written on the GPU box by tools/run_profiles.sh) -> the tracked artefacts under profiles/ (bench lines, rocprofv3 kernel stats,
PMC summaries, pmc_traffic.json with one entry per PMC'd workload).  usage: python tools/collect_profiles.py [round tag, default r06]"""
import glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r06'
src, dst = os.path.join(ROOT, 'gpurun_out', tag), os.path.join(ROOT, 'profiles')
# gpurun MERGES into gpurun_out/: delete gpurun_out/<tag> before the run, or files of earlier runs mix in
assert len(glob.glob(src + '/bench_line*.json')) >= 9 and len(glob.glob(src + '/stats_*')) >= 3 and len(glob.glob(src + '/pmc_*')) >= 3, 'incomplete ' + src
for f in sorted(glob.glob(src + '/bench_line*.json')):
    lines = [l for l in open(f).read().splitlines() if l.startswith('{')]
    assert lines, f
    open(os.path.join(dst, '%s_%s' % (tag, os.path.basename(f))), 'w').write(lines[-1] + '\n')
for d in sorted(glob.glob(src + '/stats_*')):
    f = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)
    assert len(f) == 1, d
    shutil.copy(f[0], os.path.join(dst, '%s_%s_kernel_stats.csv' % (tag, os.path.basename(d)[len('stats_'):])))
out = os.path.join(dst, 'pmc_traffic.json')
if os.path.exists(out):
    os.remove(out)
for d in sorted(glob.glob(src + '/pmc_*')):
    w = os.path.basename(d)[len('pmc_'):]
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'pmc_to_csv.py'), d, os.path.join(dst, '%s_pmc_summary_%s.csv' % (tag, w))])
    workload, dtype = w.rsplit('_', 1)
    width = 256
    if workload.startswith('w512'):
        workload, width = 'c2', 512
    if workload == 'shipped':
        width = 512
    rays = {'c2': 2048, 'c3': 8192, 'c5share': 4096, 'shipped': 32768}[workload]
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'pmc_to_json.py'), d, workload, dtype, str(width), out,
                           '%s %s: separate --pmc FETCH_SIZE / WRITE_SIZE passes of python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline [%s] (tools/run_profiles.sh)' % (tag, w, w),
                           'weak', str(rays)])
for f in sorted(glob.glob(src + '/step_timeline_*.txt')):
    shutil.copy(f, os.path.join(dst, '%s_%s' % (tag, os.path.basename(f))))
print(sorted(json.load(open(out))['entries']))
