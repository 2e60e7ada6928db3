"""Per-kernel statistics over the LAST `steps` steps of a rocprofv3 kernel trace of scratch/r06/small_step.py: every kernel's
dispatch count in the window = count_total * steps / total_steps is not known, so the window is cut by TIME: the last
`steps` step periods before the final dispatch (period = the measured ms per step).
    python3 tail_stats.py trace_results.db out.csv steps ms_per_step"""
import sqlite3, sys, collections
db, out, steps, ms = sys.argv[1], sys.argv[2], int(sys.argv[3]), float(sys.argv[4])
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
t_end = max(r[2] for r in rows)
t0 = t_end - steps * ms * 1e6
per = collections.OrderedDict()
busy = 0
for n, s, e in rows:
    if s >= t0:
        per.setdefault(n, []).append(e - s)
        busy += e - s
stats = sorted(((n, len(d), sum(d), sum(d) / len(d), min(d), max(d)) for n, d in per.items()), key=lambda r: -r[2])
with open(out, 'w') as f:
    f.write('# rocprofv3 --kernel-trace, dispatches of the last %d steps (%.3f ms per step by the host clock): %d launches per step, '
            '%.3f ms of kernel time per step\n' % (steps, ms, sum(r[1] for r in stats) // steps, busy / steps / 1e6))
    f.write('Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,NsPerStep\n')
    for n, c, t, a, lo, hi in stats:
        f.write('"%s",%d,%d,%.1f,%.2f,%d,%d,%.0f\n' % (n.replace('"', "'")[:150], c, t, a, 100.0 * t / busy, lo, hi, t / steps))
print(open(out).readline().strip())
