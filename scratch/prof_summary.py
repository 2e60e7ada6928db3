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
                f.write('"%s",%s,%.1f,%d\n' % (k.replace('(anonymous namespace)::', '').split('(')[0][:80], c, v / n, n))
def kernel_stats_timed(path, out, total_steps, skip_steps):
    """Per-kernel statistics over the dispatches of the TIMED steps only: a kernel launched c times in `total_steps` identical
    steps runs c / total_steps times per step; its first skip_steps * c / total_steps dispatches (priming + warm-up steps,
    where first-touch launches are slower) are dropped.  Kernels whose count is not a multiple of total_steps (one-off
    set-up work) keep all their dispatches and are marked with steps = 0."""
    db = sqlite3.connect(path); cur = db.cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    per = collections.OrderedDict()
    for n, s, e in rows:
        per.setdefault(n, []).append(e - s)
    stats = []
    for n, d in per.items():
        c = len(d)
        if c % total_steps == 0 and c >= total_steps:
            k = c // total_steps
            d = d[skip_steps * k:]
            steps = total_steps - skip_steps
        else:
            steps = 0
        stats.append((n, len(d), sum(d), sum(d) / len(d), min(d), max(d), steps))
    stats.sort(key=lambda r: -r[2])
    tot = sum(r[2] for r in stats if r[6]) or 1
    with open(out, 'w') as f:
        f.write('# rocprofv3 --kernel-trace, dispatches of the TIMED steps only (%d of %d steps; from %s)\n' % (total_steps - skip_steps, total_steps, path.split('/')[-1]))
        f.write('Name,Calls,TotalDurationNs,AverageNs,PercentageOfTimedSteps,MinNs,MaxNs,Steps,NsPerStep\n')
        for n, c, t, a, lo, hi, st in stats:
            f.write('"%s",%d,%d,%.1f,%.2f,%d,%d,%d,%.0f\n' % (n.replace('"', "'")[:160], c, t, a, 100.0 * t / tot if st else 0.0, lo, hi, st, t / st if st else 0))
if __name__ == '__main__':
    if sys.argv[1] == 'kernels_timed':
        kernel_stats_timed(sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])); sys.exit(0)
    mode, path, out = sys.argv[1:4]
    (kernel_stats if mode == 'kernels' else pmc_stats)(path, out)
