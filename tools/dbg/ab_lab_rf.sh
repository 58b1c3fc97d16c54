# usage (GPU box): bash tools/dbg/ab_lab_rf.sh "A B ..." ROUNDS  -> ms/step + the attention forward's in-step / probe time for tools/lab_lib.py variants
V=$1; R=$2; shift 2
for r in $(seq 1 $R); do
  for v in $V; do
    python tools/lab_lib.py bench $v --no-cpu-baseline --steps 30 --warmup 8 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); rf=d['roofline']; print('$v', round(d['ms_per_step'],3), 'fwd in-step us', round(1e3*rf['avg_launch_ms'],1), 'probe us', round(1e3*rf['probe_avg_launch_ms'],1), 'loss', d['config']['final_loss'])"
  done
done
