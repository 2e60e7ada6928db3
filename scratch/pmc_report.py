import sqlite3, sys, collections
def report(path, like=('k_sca_fwd','k_sca_bwd')):
    db = sqlite3.connect(path); cur = db.cursor()
    cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
    rows = cur.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name").fetchall()
    out = collections.defaultdict(dict)
    for k, c, v, n in rows:
        if any(l in k for l in like):
            out[k.split('<')[0].replace('void ','')][c] = (v, n)
    for k, d in out.items():
        print(k)
        for c, (v, n) in sorted(d.items()):
            print('   %-28s per-dispatch %16.1f  (n=%d)' % (c, v / n, n))
for p in sys.argv[1:]:
    print('==', p); report(p)
