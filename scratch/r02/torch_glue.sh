cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02; rm -rf gpurun_out/r02/tg
rocprofv3 --kernel-trace --stats -d gpurun_out/r02/tg -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02/tg.log 2>&1
python scratch/prof_summary.py kernels $(ls gpurun_out/r02/tg/*results.db | head -1) gpurun_out/r02/tg_stats.csv
rm -rf gpurun_out/r02/tg
python - <<'PY'
import csv
rows=[r for r in csv.reader(open('gpurun_out/r02/tg_stats.csv')) if len(r)>5 and r[0]!='Name']
tot=sum(int(r[2]) for r in rows); steps=5
cat={'gemm':0,'ours':0,'torch':0,'copy':0}
for r in rows:
    n=r[0]; t=int(r[2])
    if n.startswith('Cijk') or n.startswith('Custom'): cat['gemm']+=t
    elif 'at::native' in n or 'rocclr' in n or 'Memcpy' in n: cat['torch']+=t
    else: cat['ours']+=t
print('per step ms: total %.1f gemm %.1f ours %.1f torch %.1f' % tuple(v/1e6/steps for v in (tot,cat['gemm'],cat['ours'],cat['torch'])))
k=0
for r in rows:
    n=r[0]
    if 'at::native' in n or 'rocclr' in n:
        print('%7.2f ms/step %5s calls  %s' % (int(r[2])/1e6/steps, r[1], n[:150])); k+=1
        if k>22: break
PY
