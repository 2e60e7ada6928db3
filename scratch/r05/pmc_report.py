"""Per-kernel averages of every counter of a rocprofv3 --pmc results db (append mode):  pmc_report.py <db> <out> [name filter ...]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
like = sys.argv[3:]
rows = cur.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name").fetchall()
with open(sys.argv[2], 'a') as f:
    for k, c, v, n in sorted(rows):
        if not like or any(l in k for l in like):
            f.write('"%s",%s,%.1f,%d\n' % (k.replace('(anonymous namespace)::', '').split('(')[0][:100], c, v / n, n))
