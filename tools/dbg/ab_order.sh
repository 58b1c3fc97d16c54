A="--steps 60 --warmup 10 --no-cpu-baseline --probe-launches 0"
O=medical_tri_modal_pilot_amd.tuning
for i in 1 2 3; do
for spec in "(0,1,2)|(0,1,2)" "(2,1,0)|(0,1,2)" "(1,2,0)|(1,2,0)" "(2,1,0)|(2,1,0)"; do b=${spec%%|*}; f=${spec##*|}
  timeout -k 10 200 python tools/dbg/ab_patch.py "$O.STREAM_ISSUE_ORDER_BWD=$b" "$O.STREAM_ISSUE_ORDER_FWD=$f" -- $A 2>/dev/null | sed "s/^/bwd $b fwd $f  /"
done; done
