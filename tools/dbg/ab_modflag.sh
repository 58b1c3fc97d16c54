# A/B of a boolean module-level switch (full dotted path) in one box: bash tools/dbg/ab_modflag.sh PATH [rounds] [bench flags]
P=$1; R=${2:-2}; shift 2
A="--no-cpu-baseline --steps 30 --warmup 10 --probe-launches 0 --instep-steps 12 $@"
for r in $(seq 1 $R); do
  for m in False True; do
    echo -n "$P=$m   "; python tools/dbg/ab_patch.py "$P=$m" -- $A 2>/dev/null | tail -1 || exit 1
  done
done
