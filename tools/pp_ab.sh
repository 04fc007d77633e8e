# round 6: the product library against a side build with one engine choice switched back, IN the step, alternating runs on one box.  Build the side library
# first (CPU container), e.g.
#   python3 -c "from mvsdf_amd import build; build.build(tag='nopp', extra_flags=['-DMV_BS_PP=0'])"        # one activation tile (round 5) instead of ping-pong tiles
#   python3 -c "from mvsdf_amd import build; build.build(tag='nw8', extra_flags=['-DMV_SPHERE_NW16=0'])"   # 8 waves x 2 column tiles in k_sphere_trace instead of 16 x 1
# usage (GPU box): bash tools/pp_ab.sh [tag] [bench args]   -> gpurun_out/r06/ab_<tag>/*.json + a table on stdout  ("pp" in the table = the product library)
set -u
R=$GRAFT_REPO_ROOT
ALT=${1:-nopp}
shift || true
O=$R/gpurun_out/r06/ab_$ALT
mkdir -p $O
for i in 1 2 3; do
  python3 $R/bench.py --gpus 1 --steps 100 --warmup 30 --no-cpu-baseline "$@" > $O/pp_$i.json 2>/dev/null
  MVSDF_LIB=$R/mvsdf_amd/libmvsdf_hip_$ALT.so python3 $R/bench.py --gpus 1 --steps 100 --warmup 30 --no-cpu-baseline "$@" > $O/alt_$i.json 2>/dev/null
done
python3 - <<PY
import json
print('library | run | ms/step | kernel ms/step | per-section kernel ms')
for m in ('pp', 'alt'):
    for i in (1, 2, 3):
        d = json.loads([l for l in open('$O/%s_%d.json' % (m, i)).read().splitlines() if l.startswith('{')][-1])
        k = d['timing']['kernel_ms']
        print('%-5s | %d | %.3f | %.3f | %s' % (m, i, d['ms_per_step'], d['timing']['kernel_ms_per_step'], '  '.join('%s %.3f' % (n, k[n]) for n in sorted(k))))
PY
# per-kernel averages of the two libraries (rocprofv3 kernel trace of 60 steps each)
cd /tmp && export TMPDIR=/tmp
for m in pp alt; do
  if [ $m = alt ]; then export MVSDF_LIB=$R/mvsdf_amd/libmvsdf_hip_$ALT.so; else unset MVSDF_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -- python3 $R/bench.py --gpus 1 --steps 60 --warmup 10 --no-cpu-baseline "$@" > $O/prof_$m.log 2>&1
  f=$(find $O/prof_$m -name "*kernel_stats.csv" | head -1)
  python3 -c "
import csv, sys
print('== $m: top kernels')
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i < 7: print('  %-64s calls %4s  avg %7.1f us  total %7.2f ms' % (r['Name'][:64], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
" "$f"
  cp "$f" $O/${m}_kernel_stats.csv; rm -rf $O/prof_$m
done
