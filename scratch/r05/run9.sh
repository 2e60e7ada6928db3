cd $GRAFT_REPO_ROOT
R=gpurun_out/r05; mkdir -p $R
timeout 900 python scratch/r04/gemm_ledger.py --workload vocc_full_train --batch 64 --out $R/r05_gemm_ledger_full64.csv > $R/ledger.txt 2>&1; echo "ledger rc $?"
grep -i "SK3" $R/r05_gemm_ledger_full64.csv | cut -c1-330
echo ---
sort -t, -k10 -g -r $R/r05_gemm_ledger_full64.csv | head -25 | cut -c1-250
