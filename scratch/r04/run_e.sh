#!/bin/bash
python scratch/r04/gemm_ledger.py --workload vocc_full_train --batch 64 --micro 64 --out gpurun_out/r04_gemm_ledger_full64.csv > /dev/null 2>gpurun_out/r04_ledger_full.err; grep "other-by-op" gpurun_out/r04_gemm_ledger_full64.csv | cut -c1-190
