set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06/diag7
mkdir -p $O
B="python3 $R/bench.py --no-cpu-baseline --workload c3 --steps 40 --warmup 10"
line() { python3 -c "
import json
d=json.load(open('$1'));t=d['timing']
print('$1'.split('/')[-1], 'ms %.3f gpu %.3f kernel %.3f host_loop %.3f' % (d['ms_per_step'], t['gpu_ms_per_step'], t['kernel_ms_per_step'], t['host_ms_per_step_enqueue_loop']))"; }
$B > $O/c3_def.json 2>$O/err.txt; line $O/c3_def.json
MVSDF_DEFERRED_STEP=0 $B > $O/c3_cls.json 2>$O/err.txt; line $O/c3_cls.json
MVSDF_SPLIT_ROWS=0 $B > $O/c3_def_nosplit.json 2>$O/err.txt; line $O/c3_def_nosplit.json
MVSDF_SPLIT_ROWS=0 MVSDF_DEFERRED_STEP=0 $B > $O/c3_cls_nosplit.json 2>$O/err.txt; line $O/c3_cls_nosplit.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace_c3_def -- python3 $R/bench.py --no-cpu-baseline --workload c3 --steps 12 --warmup 4 > /dev/null 2>&1
python3 $R/tools/trace_timeline.py $O/trace_c3_def 9 > $O/timeline_c3_def.txt 2>&1
tail -45 $O/timeline_c3_def.txt
find $O -name "*agent_info.csv" -delete
