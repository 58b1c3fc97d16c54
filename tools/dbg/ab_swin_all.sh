# frozen-encoder fusions + tail split: all off vs all on, alternating, inside one gpurun call
for v in 0 1 0 1 0 1; do
  MTMP_SWIN_MLP=$v MTMP_SWIN_SPLIT_TAIL=$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 --warmup 10 --probe-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('swin fusions+split=$v', round(d['ms_per_step'],3))" || exit 1
done
