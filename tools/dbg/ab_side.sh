# A/B of ops.SIDE_AFTER_ATTN in one box: bash tools/dbg/ab_side.sh [rounds] [bench flags]
O=medical_tri_modal_pilot_amd.ops
R=${1:-2}; shift
A="--no-cpu-baseline --steps 30 --warmup 10 --probe-launches 0 --instep-steps 12 $@"
for r in $(seq 1 $R); do
  for m in False True; do
    echo -n "SIDE_AFTER_ATTN=$m   "; python tools/dbg/ab_patch.py "$O.SIDE_AFTER_ATTN=$m" -- $A 2>/dev/null | tail -1 || exit 1
  done
done
