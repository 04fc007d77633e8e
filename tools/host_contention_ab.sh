# round 6: the driver's command (20 steps, 5 warm-up) with the deferred and the classic step, on an idle host and on a host whose every hardware thread
# is taken by a busy loop, + the kernel timeline of the deferred step.  usage (GPU box): bash tools/host_contention_ab.sh
# RESULT of the one run (gpurun_out/r06/diag3, profiles/r06_driver_cmd_boxB_*): idle host -- deferred 1.520 / 1.541 ms, classic 1.591 / 1.524 (gpu_idle_frac 0.025 / 0.004).
# Busy host -- the experiment is TOO brutal to be informative: with nproc busy loops every Python statement of the benchmark takes milliseconds (host enqueue
# loop 5-65 ms per step in either mode: zero_grad alone 5-10 ms), i.e. the host is slower than the GPU on average, which no amount of run-ahead absorbs; the
# classic legs hit the call's time limit.  A milder load (a fraction of the threads) is the experiment to repeat; budget ~25 GPU-minutes for this form.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06/diag3
mkdir -p $O
B="python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
line() { python3 -c "
import json,sys
d=json.load(open('$1'))
t=d['timing']
print('$1'.split('/')[-1], 'ms_per_step %.3f gpu_ms %.3f kernel_ms %s idle %s host_loop %.3f' % (d['ms_per_step'], t['gpu_ms_per_step'], t['kernel_ms_per_step'], t['gpu_idle_frac'], t['host_ms_per_step_enqueue_loop']), t['kernel_ms'], t['host_ms'])
"; }
for i in 1 2; do $B > $O/idle_deferred_$i.json 2>$O/err.txt || tail -5 $O/err.txt; line $O/idle_deferred_$i.json; done
for i in 1 2; do MVSDF_DEFERRED_STEP=0 $B > $O/idle_classic_$i.json 2>$O/err.txt || tail -5 $O/err.txt; line $O/idle_classic_$i.json; done
# every hardware thread busy
N=$(nproc)
PIDS=""
for i in $(seq $N); do ( while :; do :; done ) & PIDS="$PIDS $!"; done
sleep 2
for i in 1 2 3; do $B > $O/busy_deferred_$i.json 2>$O/err.txt || tail -5 $O/err.txt; line $O/busy_deferred_$i.json; done
for i in 1 2 3; do MVSDF_DEFERRED_STEP=0 $B > $O/busy_classic_$i.json 2>$O/err.txt || tail -5 $O/err.txt; line $O/busy_classic_$i.json; done
kill $PIDS 2>/dev/null
wait 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_deferred -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python3 $R/tools/trace_timeline.py $O/trace_deferred 15 > $O/timeline_deferred.txt 2>&1
tail -50 $O/timeline_deferred.txt
f=$(ls $O/trace_deferred/*/*kernel_stats.csv | head -1); python3 $R/tools/kstats.py $f 45 24 > $O/kstats_deferred.txt; cat $O/kstats_deferred.txt
find $O -name "*agent_info.csv" -delete
