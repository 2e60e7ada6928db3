timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --tb=short 2>&1 | tail -6 > gpurun_out/t10.log
timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/b7_b32.json 2> gpurun_out/b7.err
timeout 600 python bench.py --steps 4 --warmup 2 --batch 64 --no-cpu-baseline > gpurun_out/b7_b64.json 2>> gpurun_out/b7.err
