set -x
python bench.py --steps 2 --warmup 1 --batch 4 --micro 4 --dtype fp32 --no-cpu-baseline > gpurun_out/b_fp32_b4.json 2> gpurun_out/b_fp32_b4.err
python bench.py --steps 3 --warmup 2 --batch 16 --micro 8 --dtype bf16 --no-cpu-baseline > gpurun_out/b_bf16_b16.json 2> gpurun_out/b_bf16_b16.err
python bench.py --steps 3 --warmup 2 --batch 32 --micro 8 --dtype bf16 --no-cpu-baseline > gpurun_out/b_bf16_b32.json 2> gpurun_out/b_bf16_b32.err
python bench.py --steps 3 --warmup 2 --batch 64 --micro 8 --dtype bf16 --no-cpu-baseline > gpurun_out/b_bf16_b64.json 2> gpurun_out/b_bf16_b64.err
tail -3 gpurun_out/b_*.err
cat gpurun_out/b_*.json
