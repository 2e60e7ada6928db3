#!/bin/bash
# N-split occupancy-MLP backward: timing-only builds without the row-view (LayerNorm) work / without the feature-view (MFMA + LDS operand) work
for abl in NONE NOROW NOFEAT; do
  echo "== $abl"
  VER_LIB=$PWD/vln-ver_amd/libver_abl_$abl.so python scratch/r02/occ_mlp_micro.py 32256000 2>&1 | tail -1
done
