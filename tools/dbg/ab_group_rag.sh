# A/B of the launch-group modes on the ragged workload in one box: bash tools/dbg/ab_group_rag.sh [rounds]
O=medical_tri_modal_pilot_amd.tuning
A="--no-cpu-baseline --steps 30 --warmup 10 --probe-launches 0 --instep-steps 0 --workload ragged"
for r in $(seq 1 ${1:-2}); do
  for m in small all none; do
    echo -n "GROUP_MODE=$m   "; python tools/dbg/ab_patch.py "$O.GROUP_MODE='$m'" -- $A 2>/dev/null | tail -1 || exit 1
  done
done
