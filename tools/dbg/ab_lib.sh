# usage: ab_lib.sh <alt .so>   -- bench.py ms/step with the in-tree library and with an alternative build, alternating
alt=$1
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 --warmup 10 --probe-launches 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), d['config']['final_loss'])"; }
for i in 1 2 3; do
  run default || exit 1
  MTMP_LIB=$alt run alt || exit 1
done
