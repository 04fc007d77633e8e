for wl in c2 c3; do
for sp in 1 0; do
for w8 in 0 1; do
echo "== $wl split=$sp w8=$w8"
MVSDF_SPLIT_GEMM=$sp MVSDF_CHAIN_W8=$w8 python bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], json.dumps(d.get('phases_ms', d.get('roofline', {}).get('step', {}))))
"
done; done; done
