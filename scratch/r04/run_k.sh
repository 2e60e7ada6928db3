#!/bin/bash
cd $GRAFT_REPO_ROOT
python scratch/r04/bench_focal.py 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_hip_ops_gpu.py tests/test_head_gpu.py -x -q -k "focal or label or occ_mlp_focal or fused" 2>&1 | tail -3
