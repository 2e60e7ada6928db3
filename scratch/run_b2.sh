timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --tb=short 2>&1 | tail -12 > gpurun_out/t6.log
timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/b2_b32.json 2> gpurun_out/b2.err
timeout 600 python bench.py --steps 3 --warmup 2 --batch 64 --no-cpu-baseline > gpurun_out/b2_b64.json 2>> gpurun_out/b2.err
