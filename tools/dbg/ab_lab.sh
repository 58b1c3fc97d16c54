# usage (GPU box): bash tools/dbg/ab_lab.sh "A B ..." ROUNDS [bench flags]  -> ms/step of tools/lab_lib.py variants, alternating
V=$1; R=$2; shift 2
for r in $(seq 1 $R); do
  for v in $V; do
    python tools/lab_lib.py bench $v --no-cpu-baseline --probe-launches 0 --instep-steps 0 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3))"
  done
done
