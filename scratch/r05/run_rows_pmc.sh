cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=gpurun_out/rp; mkdir -p $R
timeout 300 python scratch/r05/rows_bench.py 64 10
for C in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"; do
timeout 300 rocprofv3 --kernel-trace --pmc $C -d $R/p -o pmc -- python3 scratch/r05/rows_bench.py 64 2 > /dev/null 2> $R/err.txt
python - <<PY
import sqlite3
db = sqlite3.connect('$R/p/pmc_results.db'); cur = db.cursor()
rows = cur.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name").fetchall()
for k, c, v, n in sorted(rows):
    if 'k_lattice_rows' in k or 'k_lattice_transpose' in k or 'k_run_copy' in k:
        print('%-60s %-22s %14.1f per dispatch (%d)' % (k.split('(')[0][-60:], c, v / n, n))
PY
rm -rf $R/p
done
