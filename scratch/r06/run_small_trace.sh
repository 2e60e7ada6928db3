cd /tmp && export TMPDIR=/tmp
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
cd $GRAFT_REPO_ROOT
R=gpurun_out/r06; mkdir -p $R
for B in ${BATCHES:-1 8}; do
  timeout 600 rocprofv3 --kernel-trace -d $R/trace_b$B -o trace -- python3 scratch/r06/small_step.py $B 20 1 > $R/small_b$B.txt 2> $R/small_b$B.err
  MS=$(grep -o '[0-9.]* ms per step' $R/small_b$B.txt | cut -d' ' -f1)
  python3 scratch/r06/tail_stats.py $(ls $R/trace_b$B/*/trace_results.db $R/trace_b$B/trace_results.db 2>/dev/null | head -1) $R/b${B}_kernel_stats.csv 20 $MS
  rm -rf $R/trace_b$B
  cat $R/small_b$B.txt
done
