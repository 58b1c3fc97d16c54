# HBM traffic of the frozen encoder's kernels (fused and un-fused stage 1-2 launches): two separate --pmc passes each
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in 1 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    MTMP_SWIN_MLP=$mode PARTS=1 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_swin_${mode}_$c -o x --output-format csv -- python3 $R/tools/dbg/swin_split.py > /dev/null 2>&1
    python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_swin_${mode}_$c/x_counter_collection.csv --json $R/gpurun_out/pmc_swin_${mode}_$c.json > /dev/null
    echo done $mode $c
  done
done
