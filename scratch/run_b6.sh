timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --tb=short 2>&1 | tail -6 > gpurun_out/t9.log
timeout 600 python bench.py --steps 3 --warmup 2 --workload vocc_full_train --batch 16 > gpurun_out/b6_full.json 2> gpurun_out/b6.err
timeout 600 python bench.py --steps 3 --warmup 2 --workload c2_single_scale_fwd --batch 4 --micro 4 --dtype fp32 > gpurun_out/b6_c2.json 2>> gpurun_out/b6.err
