# round 6: a SLOW HOST in a controlled form -- bench.py --host-delay-us D busy-waits D microseconds after each of the six calls of a step (6 D per step on top
# of the ~0.4 ms of real enqueueing) -- with the deferred step (default) and the classic step (MVSDF_DEFERRED_STEP=0), the driver's protocol (20 steps, 5 warm-up).
# usage (GPU box): bash tools/host_delay_ab.sh   -> gpurun_out/r06/host_delay/*.json + a table on stdout
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06/host_delay
mkdir -p $O
for D in 0 50 100 150 200 300; do
  python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --host-delay-us $D > $O/deferred_$D.json 2>/dev/null
  MVSDF_DEFERRED_STEP=0 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --host-delay-us $D > $O/classic_$D.json 2>/dev/null
done
python3 - <<PY
import json
print('host delay per call (us) | deferred: ms/step, gpu idle | classic: ms/step, gpu idle')
for D in (0, 50, 100, 150, 200, 300):
    r = []
    for m in ('deferred', 'classic'):
        d = json.loads([l for l in open('$O/%s_%d.json' % (m, D)).read().splitlines() if l.startswith('{')][-1])
        r.append('%.3f  %.3f' % (d['ms_per_step'], d['timing']['gpu_idle_frac']))
    print('%6d                   | %s           | %s' % (D, r[0], r[1]))
PY
