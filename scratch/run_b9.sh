timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --tb=short 2>&1 | tail -8 > gpurun_out/t13.log
timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/b9_b32.json 2> gpurun_out/b9.err
