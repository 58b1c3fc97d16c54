python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep '"metric"' | cut -c1-330
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
