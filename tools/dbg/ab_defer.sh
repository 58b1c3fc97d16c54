A="--steps 60 --warmup 10 --no-cpu-baseline --probe-launches 0"
for i in 1 2 3; do
  timeout -k 10 200 python tools/dbg/ab_patch.py "medical_tri_modal_pilot_amd.ops.DEFER_REDUCTIONS=True" -- $A 2>&1 | tail -1 | sed "s/^/one launch per layer  /"
  timeout -k 10 200 python tools/dbg/ab_patch.py "medical_tri_modal_pilot_amd.ops.DEFER_REDUCTIONS=False" -- $A 2>&1 | tail -1 | sed "s/^/seven launches        /"
done
