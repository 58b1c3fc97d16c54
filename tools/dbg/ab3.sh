# usage: ab3.sh <name=lib.so> ...   -- bench.py ms/step with the in-tree library and with alternative builds, alternating (3 rounds)
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 --warmup 10 --probe-launches 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), d['config']['final_loss'])"; }
for i in 1 2 3; do
  (unset MTMP_LIB; run default) || exit 1
  for a in "$@"; do MTMP_LIB=${a#*=} run ${a%%=*} || exit 1; done
done
