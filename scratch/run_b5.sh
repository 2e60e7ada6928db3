VER_BIAS_TAPS=mv timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/b5_mv.json 2> gpurun_out/b5.err
VER_BIAS_TAPS=old timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/b5_old.json 2>> gpurun_out/b5.err
VER_BIAS_TAPS=mv timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/b5_mv2.json 2>> gpurun_out/b5.err
VER_BIAS_TAPS=old timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/b5_old2.json 2>> gpurun_out/b5.err
