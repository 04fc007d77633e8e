# round 6: the driver's command (20 steps, 5 warm-up) with the deferred and the classic step, on an idle host and on a host whose every core is taken
# by a busy loop (what a shared node does to a step that waits for the GPU in its middle), + the kernel timeline of the deferred step
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
