"""The frozen image encoder's stage-3 / stage-4 projections alone (half batch = 32 images, full batch = 64): mtmp_gemm_nt
against the library GEMM (torch -> hipBLASLt) on the same data, 20 launches of one shape captured in a hipGraph and replayed.

    python tools/dbg/small_gemm_bench.py [--only qkv3,fc2_3]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from medical_tri_modal_pilot_amd import ops  # noqa: E402

DEV = torch.device("cuda", 0)
BF = torch.bfloat16
SHAPES = {          # name: (rows per image, N, K, epilogue)
    "qkv3": (196, 1152, 384, "bias"), "proj3": (196, 384, 384, "res"), "fc1_3": (196, 1536, 384, "gelu"), "fc2_3": (196, 384, 1536, "res"),
    "qkv4": (49, 2304, 768, "bias"), "proj4": (49, 768, 768, "res"), "fc1_4": (49, 3072, 768, "gelu"), "fc2_4": (49, 768, 3072, "res"),
}


def graph_time(fn, reps=20, replays=10):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=torch.cuda.Stream(device=DEV)):
        for _ in range(reps):
            fn()
    for _ in range(2):
        g.replay()
    ts = []
    for _ in range(replays):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    only = [s for s in a.only.split(",") if s]
    g = torch.Generator(device=DEV).manual_seed(0)
    R = lambda *s: torch.randn(*s, generator=g, device=DEV, dtype=torch.float32).to(BF)
    for name, (rpi, n, k, epi) in SHAPES.items():
        if only and name not in only:
            continue
        for imgs in (32, 64):
            m = imgs * rpi
            x, w, b, r = R(m, k), R(n, k) * 0.05, torch.zeros(n, device=DEV), R(m, n)
            rs = torch.ones(imgs, device=DEV)
            if epi == "bias":
                ours = lambda: ops.gemm_nt(x, w, b)
                blas = lambda: torch.nn.functional.linear(x, w, b.to(BF))
            elif epi == "gelu":
                ours = lambda: ops.gemm_nt(x, w, b, act="gelu")
                blas = lambda: torch.nn.functional.gelu(torch.nn.functional.linear(x, w, b.to(BF)))
            else:
                ours = lambda: ops.gemm_nt(x, w, b, res2d=r, row_scale=rs, rows_per_scale=rpi)
                blas = lambda: torch.addmm(r, x, w.t())
            fl = 2.0 * m * n * k
            to, tb = graph_time(ours), graph_time(blas)
            print(f"{name:6s} M={m:6d} N={n:5d} K={k:5d}: mtmp {to:6.1f} us {fl / to / 1e6:6.0f} TF/s | library {tb:6.1f} us {fl / tb / 1e6:6.0f} TF/s",
                  flush=True)


if __name__ == "__main__":
    main()
