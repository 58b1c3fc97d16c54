A="--steps 60 --warmup 10 --no-cpu-baseline --probe-launches 0"
for i in 1 2 3; do
  timeout -k 10 200 python tools/dbg/ab_patch.py "medical_tri_modal_pilot_amd.tuning.LATE_REDUCTIONS=True" -- $A 2>&1 | tail -1 | sed "s/^/late      /"
  timeout -k 10 200 python tools/dbg/ab_patch.py "medical_tri_modal_pilot_amd.tuning.LATE_REDUCTIONS=False" -- $A 2>&1 | tail -1 | sed "s/^/in place  /"
done
