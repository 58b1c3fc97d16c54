# SQ counters of the attention kernels at the config-2 shape: bash tools/pmc_attn.sh <tag> [bwd]
# (counter passes only -- never combined with tracing; results in gpurun_out/<tag>_attn_sq_<i>.json)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$1; BWD=$2
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $R/gpurun_out/pmc_${T}_$i -o x --output-format csv -- python3 $R/tools/dbg/attn_only.py 5 bounded $BWD > $R/gpurun_out/pmc_${T}_$i.log 2>&1 \
    && python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${T}_$i/x_counter_collection.csv --match attn --json $R/gpurun_out/${T}_attn_sq_$i.json > /dev/null 2>&1
  echo done set $i
done
