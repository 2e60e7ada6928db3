cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|GRBM_[A-Z_]*\|TCC_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|FETCH_SIZE\|WRITE_SIZE\|LDSBankConflict\|MemUnitStalled" | sort -u | tr '\n' ' ' > gpurun_out/counters.txt
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d gpurun_out/pmc1 -o p1 -- python3 scratch/bench_gather.py 64 > gpurun_out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAVES -d gpurun_out/pmc2 -o p2 -- python3 scratch/bench_gather.py 64 > gpurun_out/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc3 -o p3 -- python3 scratch/bench_gather.py 64 > gpurun_out/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc4 -o p4 -- python3 scratch/bench_gather.py 64 > gpurun_out/pmc4.log 2>&1
ls gpurun_out/pmc*/
