#!/bin/bash
# fused occupancy MLP backward: N-split kernel (8 / 4 waves) against the row-split kernel + host GEMM, 32 M rows
for m in "1 8" "1 4" "0 8"; do
  set -- $m
  echo "== VER_OCC_MLP_BWD_FUSED=$1 VER_OCC_MLP_NS_WAVES=$2"
  VER_OCC_MLP_BWD_FUSED=$1 VER_OCC_MLP_NS_WAVES=$2 python scratch/r02/occ_mlp_micro.py 32256000 2>&1 | tail -1
done
