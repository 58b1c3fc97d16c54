# SQ counters of the attention kernels (tools/dbg/attn_only.py 5 bwd) and the weight-gradient GEMM (bench_kernels --only gemm_tn)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $R/gpurun_out/pmc_hv_attn_$i -o x --output-format csv -- python3 $R/tools/dbg/attn_only.py 5 bwd > $R/gpurun_out/pmc_hv_attn_$i.log 2>&1 && python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_hv_attn_$i/x_counter_collection.csv --match attn --json $R/gpurun_out/pmc_hv_attn_$i.json > /dev/null 2>&1
  rocprofv3 --pmc $set -d $R/gpurun_out/pmc_hv_tn_$i -o x --output-format csv -- python3 $R/tools/bench_kernels.py --only gemm_tn --rounds 3 > $R/gpurun_out/pmc_hv_tn_$i.log 2>&1 && python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_hv_tn_$i/x_counter_collection.csv --match gemm_tn --json $R/gpurun_out/pmc_hv_tn_$i.json > /dev/null 2>&1
  echo done set $i
done
