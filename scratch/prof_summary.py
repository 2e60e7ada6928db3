"""Turn a rocprofv3 results .db (kernel trace and/or PMC) into small text summaries."""
import sqlite3, sys, collections
def kernel_stats(path, out):
    db = sqlite3.connect(path); cur = db.cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open(out, 'w') as f:
        f.write('# rocprofv3 --kernel-trace --stats summary (from %s)\n' % path.split('/')[-1])
        f.write('Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n')
        for r in rows:
            f.write('"%s",%d,%d,%.1f,%.2f,%d,%d\n' % (r[0].replace('"', "'")[:160], r[1], r[2], r[3], 100.0 * r[2] / tot, r[4], r[5]))
def pmc_stats(path, out, like=('k_sca', 'k_project', 'k_build', 'k_zero', 'k_msda', 'k_lattice', 'k_occ_mlp', 'k_focal')):
    db = sqlite3.connect(path); cur = db.cursor()
    rows = cur.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name").fetchall()
    with open(out, 'a') as f:
        f.write('# rocprofv3 --pmc per-dispatch averages (from %s)\n' % path.split('/')[-1])
        f.write('Kernel,Counter,PerDispatch,Dispatches\n')
        for k, c, v, n in sorted(rows):
            if any(l in k for l in like):
                f.write('"%s",%s,%.1f,%d\n' % (k.split('(')[0][:80], c, v / n, n))
if __name__ == '__main__':
    mode, path, out = sys.argv[1:4]
    (kernel_stats if mode == 'kernels' else pmc_stats)(path, out)
