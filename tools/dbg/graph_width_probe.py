"""Does a captured graph with W parallel branches survive capture + replay, over and over, in one process?
(ROCm 7.2 / torch 2.10: hip::Graph::UpdateStreams hands the graph's branches the exec's own streams, skipping those that share
the launch stream's queue, with no bound check -- tools/dbg notes in DESIGN section 3.)  One child process per width.

    python tools/dbg/graph_width_probe.py            # widths 2..8, 150 graphs each, kept alive
    python tools/dbg/graph_width_probe.py child W N KEEP
"""
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    W, N, keep = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dev = torch.device("cuda", 0)
    x = torch.zeros(W, 1 << 16, device=dev)
    alive = []
    for it in range(N):
        cap = torch.cuda.Stream(device=dev)
        sides = [torch.cuda.Stream(device=dev) for _ in range(W - 1)]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
            cur = torch.cuda.current_stream()
            x[0].add_(1.0)
            for k, s in enumerate(sides):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    for _ in range(3):
                        x[k + 1].add_(1.0)
            for _ in range(3):
                x[0].add_(1.0)
            for s in sides:
                cur.wait_stream(s)
        if keep:
            alive.append(g)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        if it % 25 == 0:
            print(f"W={W} graph {it} ok", flush=True)
    print(f"W={W}: {N} graphs captured and replayed", flush=True)
    sys.exit(0)

keep = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for W in (2, 3, 4, 5, 6, 8):
    r = subprocess.run([sys.executable, __file__, "child", str(W), "150", str(keep)], capture_output=True, text=True)
    last = (r.stdout.strip().splitlines() or ["-"])[-1]
    print(f"width {W} keep={keep}: rc={r.returncode}  {last}", flush=True)
