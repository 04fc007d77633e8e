"""Print a rocprofv3 kernel_stats.csv as a short table (per-step microseconds when --steps is given).  python tools/kstats.py <csv> [steps]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    per = (' per step %8.1f us' % (float(r['TotalDurationNs']) / 1e3 / steps)) if steps else ''
    print('%-64s calls %5s avg %9.1f us%s %6.2f%%' % (r['Name'][:64], r['Calls'], float(r['AverageNs']) / 1e3, per, float(r['Percentage'])))
print('total %.1f us%s' % (tot / 1e3, (' = %.1f us per step' % (tot / 1e3 / steps)) if steps else ''))
