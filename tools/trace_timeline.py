"""One training step of a rocprofv3 --kernel-trace run of bench.py as a timeline: start (us from the step's first kernel), duration, the gap to the
end of everything before it, the HIP queue, the kernel.  The step is taken from the TIMED region of bench.py (the 20 steps after it carry HIP
events around the tracer's launches, i.e. ~6 us bubbles the headline loop does not have).
usage: python tools/trace_timeline.py <rocprof output dir> [step index, default 12]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_step_prologue' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 12
a, b = idx[k], idx[k + 1]
t0 = int(rows[a]['Start_Timestamp'])
prev_end, gaps, busy = None, 0.0, 0.0
for r in rows[a:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    if r is not rows[b]:
        gaps += max(gap, 0.0)
        busy += (e - s) / 1e3
    print('%8.1f %8.1f  gap %7.1f  q%s  %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, r.get('Queue_Id', '?'), r['Kernel_Name'].split('(')[0][:70]))
    prev_end = max(prev_end or 0, e)
print('step period %.1f us, %d launches, idle between kernels %.1f us' % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3, b - a, gaps))
