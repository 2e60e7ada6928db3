#!/bin/bash
# GPU box: slot timeline + team ablations of k_occ_mlp_bwd_ws (libraries built in the container travel with the snapshot:
#   python vln-ver_amd/csrc/build.py -DVER_WS_TIMELINE --lib=scratch/r06/lib_ws_timeline.so   [+ -DVER_WS_TIMELINE_FINE -> lib_ws_timeline_fine.so]
#   python vln-ver_amd/csrc/build.py -DVER_WS_ABL_NOROW --lib=scratch/r06/lib_VER_WS_ABL_NOROW.so   (NOFEAT, NOSUMS likewise))
mkdir -p gpurun_out/r06
out=gpurun_out/r06/ws_timeline.txt
: > $out
python scratch/r06/ws_timeline.py >> $out 2>&1
for lib in lib_ws_timeline lib_ws_timeline_fine lib_VER_WS_ABL_NOROW lib_VER_WS_ABL_NOFEAT lib_VER_WS_ABL_NOSUMS; do
  [ -f scratch/r06/$lib.so ] && VER_LIB=scratch/r06/$lib.so python scratch/r06/ws_timeline.py >> $out 2>&1
done
grep -v "amdgpu.ids\|Warn" $out
