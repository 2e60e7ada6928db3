#!/bin/bash
# where the forward gather's time goes: timing-only ablation builds (results are wrong by construction)
for abl in NONE NOATOMIC NOSTORE NOCONV NOPOINTS NODMA DUMMYOUT; do
  lib=vln-ver_amd/libver_abl_$abl.so
  [ -f $lib ] || continue
  echo "== $abl"
  VER_LIB=$PWD/$lib VER_BENCH_RING=1 python scratch/bench_gather.py 192 4x15x15 bf16 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['fwd_us'])"
done
