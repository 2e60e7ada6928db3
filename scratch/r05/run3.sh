cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=gpurun_out/r05; mkdir -p $R
timeout 900 python scratch/r05/wgrad_bench.py check sweep > $R/wgrad3.txt 2>&1; echo "rc $?"
grep -c OK $R/wgrad3.txt; grep "FAIL\|Error\|error" $R/wgrad3.txt | head; grep "ver_wgrad_tn\|library" $R/wgrad3.txt
rm -f $R/wgrad_pmc.csv
for F in 0 12 -1; do
  echo "# flags $F" >> $R/wgrad_pmc.csv
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $R/p1 -o pmc -- python3 scratch/r05/wgrad_prof.py $F 8 3 > /dev/null 2> $R/p1.err; echo "pmc1 $F $?"
  python scratch/r05/pmc_report.py $R/p1/pmc_results.db $R/wgrad_pmc.csv k_wgrad Cijk; rm -rf $R/p1
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU -d $R/p2 -o pmc -- python3 scratch/r05/wgrad_prof.py $F 8 3 > /dev/null 2> $R/p2.err; echo "pmc2 $F $?"
  python scratch/r05/pmc_report.py $R/p2/pmc_results.db $R/wgrad_pmc.csv k_wgrad Cijk; rm -rf $R/p2
done
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/p3 -o pmc -- python3 scratch/r05/wgrad_prof.py 0 8 3 > /dev/null 2> $R/p3.err; echo "pmc3 $?"
python scratch/r05/pmc_report.py $R/p3/pmc_results.db $R/wgrad_pmc.csv k_wgrad Cijk; rm -rf $R/p3
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $R/p4 -o pmc -- python3 scratch/r05/wgrad_prof.py 0 8 3 > /dev/null 2> $R/p4.err; echo "pmc4 $?"
python scratch/r05/pmc_report.py $R/p4/pmc_results.db $R/wgrad_pmc.csv k_wgrad Cijk; rm -rf $R/p4
cat $R/wgrad_pmc.csv
