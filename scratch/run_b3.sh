timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --tb=short 2>&1 | tail -12 > gpurun_out/t7.log
timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/b3_b32.json 2> gpurun_out/b3.err
