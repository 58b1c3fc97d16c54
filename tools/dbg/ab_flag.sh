# A/B of a boolean module-level switch of ops in one box: bash tools/dbg/ab_flag.sh NAME [rounds] [bench flags]
O=medical_tri_modal_pilot_amd.tuning
N=$1; R=${2:-2}; shift 2
A="--no-cpu-baseline --steps 30 --warmup 10 --probe-launches 0 --instep-steps 0 $@"
for r in $(seq 1 $R); do
  for m in False True; do
    echo -n "$N=$m   "; python tools/dbg/ab_patch.py "$O.$N=$m" -- $A 2>/dev/null | tail -1 || exit 1
  done
done
