# usage (GPU box): bash tools/dbg/ab_round.sh OTHER_TREE ROUNDS [bench flags]  -> ms/step of this tree and of another checkout (its own build), alternating
O=$1; R=$2; shift 2
run() { d=$1; n=$2; shift 2; (cd $d && timeout -k 10 300 python bench.py --no-cpu-baseline --steps 40 --warmup 10 --probe-launches 0 --instep-steps 0 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', round(d['ms_per_step'],3))"); }
for r in $(seq 1 $R); do run . this "$@"; run $O other "$@"; done
