# SQ counters of the ln_gemm launches of tools/bench_kernels.py --only ln_gemm (counters only, no other tracing)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $set | cut -d" " -f1)
  rocprofv3 --pmc $set -d $R/gpurun_out/pmc_ln_$tag -o x --output-format csv -- python3 $R/tools/bench_kernels.py --only ln_gemm --rounds 3 > $R/gpurun_out/pmc_ln_$tag.log 2>&1 || { echo "set failed: $set"; tail -3 $R/gpurun_out/pmc_ln_$tag.log; continue; }
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_ln_$tag/x_counter_collection.csv --match ln_gemm --json $R/gpurun_out/pmc_ln_$tag.json > /dev/null 2>&1
  echo done $tag
done
