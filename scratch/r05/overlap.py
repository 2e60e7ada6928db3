"""How much of a traced run's kernel time overlaps across HIP streams / queues.  usage: overlap.py trace_results.db"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
print('columns:', cols)
qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
sel = "select start, end, name%s from kernels order by start" % ((', ' + qcol) if qcol else '')
rows = cur.execute(sel).fetchall()
print('dispatches', len(rows))
# last 40 % of the run = the timed steps
t0, t1 = rows[0][0], rows[-1][1]
cut = t0 + 0.6 * (t1 - t0)
rows = [r for r in rows if r[0] >= cut]
span = rows[-1][1] - rows[0][0]
tot = sum(r[1] - r[0] for r in rows)
busy = 0; cur_s = cur_e = None
for s, e, *_ in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print('span %.1f ms   sum of durations %.1f ms   union busy %.1f ms   idle %.1f ms   overlapped %.1f ms' % (span / 1e6, tot / 1e6, busy / 1e6, (span - busy) / 1e6, (tot - busy) / 1e6))
if qcol:
    per = collections.Counter(); cnt = collections.Counter()
    for s, e, n, q in rows:
        per[q] += e - s; cnt[q] += 1
    for q in per:
        print('  %s %s: %d dispatches, %.1f ms' % (qcol, q, cnt[q], per[q] / 1e6))
