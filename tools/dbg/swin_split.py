"""Frozen Swin-T forward of the bench batch (64 x 224 x 224) as ONE chain vs 2 / 4 sub-batches on separate HIP streams,
each variant captured in a hipGraph (as in the training step) and timed over 30 replays with HIP events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from medical_tri_modal_pilot_amd.builder.models import get_model

dev = torch.device("cuda", 0)
args = bench.make_args("full", "bf16", 0.1, 1, 0, False)
torch.manual_seed(0)
enc = get_model(args)(args).to(dev).img_encoder
img = torch.randn(64, 1, 224, 224, device=dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]


def run(parts):
    cur = torch.cuda.current_stream()
    outs = []
    for k, ch in enumerate(img.chunk(parts)):
        s = streams[k]
        s.wait_stream(cur)
        with torch.cuda.stream(s), torch.no_grad():
            outs.append(enc(ch, tail_streams=(streams[2], streams[3])) if os.environ.get('AUX') and parts == 1 else enc(ch))
    for s in streams:
        cur.wait_stream(s)
    return torch.cat(outs) if parts > 1 else outs[0]


for parts in [int(v) for v in os.environ.get("PARTS", "1,2,4,1,2,4").split(",")]:
    for _ in range(2):
        run(parts)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    cs = torch.cuda.Stream(device=dev)
    with torch.cuda.graph(g, stream=cs):
        out = run(parts)
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"parts={parts}: {e0.elapsed_time(e1) / 30 * 1e3:.0f} us per forward", flush=True)
