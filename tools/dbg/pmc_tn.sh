#!/bin/bash
# Diagnostic: clock, matrix-pipe and LDS counters of the weight-gradient kernel (three rocprofv3 --pmc passes over tools/bench_kernels.py).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_tn; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for c in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $c -d $O/p$i -o x --output-format csv -- python3 $R/tools/bench_kernels.py --only gemm_tn --rounds 3 > $O/p$i.log 2>&1 || { tail -5 $O/p$i.log; exit 1; }
  python3 $R/tools/pmc_summary.py $O/p$i/x_counter_collection.csv --match gemm_tn_dma > $O/p$i.txt 2>&1
  cat $O/p$i.txt
done
