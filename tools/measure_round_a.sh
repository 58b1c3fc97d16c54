# first half of tools/measure_round.sh (a gpurun call is limited to 20 minutes): GPU tests, bench.py with the CPU baseline, forced-DDP run
set -o pipefail
T=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/${T}_tests.log 2>&1 || { tail -5 $O/${T}_tests.log; exit 1; }
tail -1 $O/${T}_tests.log
cp $O/parity_report.json $O/${T}_parity_report.json; cp $O/gpu_test_memory.txt $O/${T}_gpu_test_memory.txt 2>/dev/null
python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err || { tail -5 $O/${T}_bench.err; exit 1; }
tail -1 $O/${T}_bench.json | cut -c1-300
python bench.py --no-cpu-baseline --force-ddp > $O/${T}_bench_ddp1.json 2> $O/${T}_bench_ddp1.err || { tail -5 $O/${T}_bench_ddp1.err; exit 1; }
tail -1 $O/${T}_bench_ddp1.json | cut -c1-200
