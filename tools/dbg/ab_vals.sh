# A/B of one module attribute over a list of values in one box: bash tools/dbg/ab_vals.sh PATH "v1;v2;..." [rounds] [bench flags]
P=$1; IFS=';' read -ra VALS <<< "$2"; R=${3:-2}; shift 3
A="--no-cpu-baseline --steps 30 --warmup 10 --probe-launches 0 --instep-steps 0 $@"
for r in $(seq 1 $R); do
  for v in "${VALS[@]}"; do
    echo -n "$P=$v   "; python tools/dbg/ab_patch.py "$P=$v" -- $A 2>/dev/null | tail -1 || exit 1
  done
done
