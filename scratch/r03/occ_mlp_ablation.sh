#!/bin/bash
# wave-specialised occupancy-MLP backward: timing-only builds in which the row team / the feature team only keeps the barriers
for abl in NONE NOROW NOFEAT; do
  echo "== $abl"
  VER_LIB=$PWD/vln-ver_amd/libver_abl_$abl.so python scratch/r02/occ_mlp_micro.py 32256000 2>&1 | tail -1
done
