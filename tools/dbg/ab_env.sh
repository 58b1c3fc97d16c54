# usage: ab_env.sh VAR v1 v2 ...   -- bench.py ms/step for each value of one environment switch, inside ONE gpurun call
var=$1; shift
for v in "$@"; do
  env $var=$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 --warmup 10 --probe-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', round(d['ms_per_step'],3))" || exit 1
done
