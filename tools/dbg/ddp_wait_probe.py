"""Does Work.wait() of a one-rank RCCL all_reduce block the HOST until the collective (and what it is ordered behind) has run?
    python tools/dbg/ddp_wait_probe.py"""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
x = torch.randn(8192, 8192, device=dev)
g = torch.randn(12_000_000, device=dev)
side = torch.cuda.Stream()
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        y = x @ x                      # ~20 x 1.1 TFLOP of queued work
    t1 = time.perf_counter()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        w = dist.all_reduce(g, async_op=True)
    t2 = time.perf_counter()
    w.wait()
    t3 = time.perf_counter()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"enqueue matmuls {1e3*(t1-t0):.2f} ms | all_reduce call {1e3*(t2-t1):.2f} ms | work.wait() {1e3*(t3-t2):.2f} ms | drain {1e3*(t4-t3):.2f} ms")
dist.destroy_process_group()
