"""Per-queue busy segments of one step of a traced run.  usage: timeline.py trace_results.db [step_from_end]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = cur.execute("select start, end, name, queue_id from kernels order by start").fetchall()
# step boundaries: first dispatch of the fused AdamW kernel group of every step (9+ launches in a row)
opt = [i for i, r in enumerate(rows) if 'FusedOptimizerTensorListMetadata' in r[2]]
ends = [opt[i] for i in range(len(opt)) if i + 1 == len(opt) or opt[i + 1] - opt[i] > 50]
print('steps found', len(ends))
lo, hi = ends[-1 - back] + 1, ends[-back] + 1
step = rows[lo:hi]
t0 = step[0][0]
print('step: %d dispatches, %.2f ms' % (len(step), (step[-1][1] - t0) / 1e6))
queues = sorted({r[3] for r in step})
for q in queues:
    segs = []
    for s, e, n, qq in step:
        if qq != q: continue
        if segs and s - segs[-1][1] < 200e3:
            segs[-1][1] = max(segs[-1][1], e); segs[-1][2] += 1; segs[-1][3] += e - s
        else:
            segs.append([s, e, 1, e - s, n])
    print('queue', q, ': %d dispatches, busy %.2f ms' % (sum(x[2] for x in segs), sum(x[3] for x in segs) / 1e6))
    for s, e, c, b, n in segs:
        if e - s > 0.3e6 or c > 20:
            print('   %8.2f -> %8.2f ms  %5d launches  busy %7.2f ms   first: %s' % ((s - t0) / 1e6, (e - t0) / 1e6, c, b / 1e6, n[:60]))

import collections
cnt = collections.Counter(); dur = collections.Counter()
for s_, e_, n, q in step:
    cnt[n] += 1; dur[n] += e_ - s_
print('--- kernels of the step by launch count')
for n, c in cnt.most_common(45):
    print('%5d  %8.3f ms  %s' % (c, dur[n] / 1e6, n[:150]))
