for S in 1 5 15 1 5; do
VER_L0_SPLIT=$S timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/b10_s$S.json 2> gpurun_out/b10.err
python -c "
import json
d=json.loads(open('gpurun_out/b10_s$S.json').read().strip().split('\n')[-1]); print('split $S', d['value'], d['ms_per_step'])
" >> gpurun_out/b10.log
done
