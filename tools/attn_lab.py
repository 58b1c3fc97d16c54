"""Diagnostic bench of patched copies of csrc/attention.hip (timing experiments; patched builds may compute WRONG results
by design and never ship).  Patches live here, not in the product source.

    python tools/attn_lab.py build            # in the build container: tools/dbg/_lab/libattn_<variant>.so
    python tools/attn_lab.py run [fwd|bwd]    # on the GPU box: interleaved rounds, median / min per variant
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "medical_tri_modal_pilot_amd", "csrc")
LAB = os.path.join(ROOT, "tools", "dbg", "_lab")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result", "-mllvm", "-amdgpu-mfma-vgpr-form",
         "-fno-slp-vectorize", "-shared", "-I", CSRC]

# name -> list of (old, new) textual replacements on attention.hip; every `old` must occur exactly once
VARIANTS = {
    "base": [],
}
try:
    sys.path.insert(0, os.path.join(ROOT, "tools", "dbg"))
    from attn_lab_variants import VARIANTS as _V          # scratch file with the experiment of the day
    VARIANTS.update(_V)
except ImportError:
    pass


def build():
    os.makedirs(LAB, exist_ok=True)
    src = open(os.path.join(CSRC, "attention.hip")).read()
    procs = []
    for name, reps in VARIANTS.items():
        s = src
        if reps and isinstance(reps[0], str):                # ("file.hip", (old, new), ...): another source file as the starting point
            s, reps = open(os.path.join(LAB, reps[0])).read(), reps[1:]
        for old, new in reps:
            assert s.count(old) == 1, f"variant {name}: pattern occurs {s.count(old)} times: {old[:60]!r}"
            s = s.replace(old, new)
        f = os.path.join(LAB, f"attn_{name}.hip")
        open(f, "w").write(s)
        out = os.path.join(LAB, f"libattn_{name}.so")
        procs.append((name, subprocess.Popen(["/opt/rocm/bin/hipcc", *FLAGS, f, os.path.join(CSRC, "error.cpp"), "-o", out],
                                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for name, p in procs:
        o, _ = p.communicate()
        print(name, "ok" if p.returncode == 0 else "FAILED\n" + o[-3000:])


def run(what):
    import torch
    B, N = 64, 1005
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B, N, 768, device="cuda", generator=g).bfloat16()
    res = torch.randn(B, N, 256, device="cuda", generator=g).bfloat16()
    do = torch.randn(B, N, 256, device="cuda", generator=g).bfloat16()
    kv = torch.full((B,), N, dtype=torch.int32, device="cuda")
    o = torch.empty(B, N, 256, device="cuda", dtype=torch.bfloat16)
    o_res = torch.empty_like(o)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(B * 4 * N, device="cuda")
    lse = torch.empty(B, 4, N, device="cuda")
    kn = torch.empty((B * N + 31) // 32, 4, device="cuda")
    P = lambda t, off=0: ctypes.c_void_p(t.data_ptr() + off)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    libs = {}
    for name in VARIANTS:
        f = os.path.join(LAB, f"libattn_{name}.so")
        if os.path.exists(f):
            libs[name] = ctypes.CDLL(f)
    first = next(iter(libs.values()))
    first.mtmp_key_norms(1, P(qkv, 512), P(kn), ctypes.c_longlong(B * N), 4, 768, st)

    def fwd(L):
        rc = L.mtmp_attn_fwd(1, P(qkv), P(qkv, 512), P(qkv, 1024), P(o), P(res), P(o_res), P(lse), P(kv), P(kn), B, N, 4, 768, 256,
                             ctypes.c_float(0.125), st)
        assert rc == 0

    def bwd(L):
        rc = L.mtmp_attn_bwd(1, P(qkv), P(qkv, 512), P(qkv, 1024), P(o), P(do), P(lse), P(kv), P(dqkv), P(dqkv, 512), P(dqkv, 1024),
                             P(delta), B, N, 4, 768, 256, 256, 768, ctypes.c_float(0.125), st)
        assert rc == 0
    fn = fwd if what == "fwd" else bwd
    fwd(first)
    torch.cuda.synchronize()
    times = {n: [] for n in libs}
    for rnd in range(12):
        for n, L in libs.items():
            fn(L)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn(L)
            e1.record()
            torch.cuda.synchronize()
            times[n].append(e0.elapsed_time(e1) / 5 * 1e3)
    for n, ts in times.items():
        ts = sorted(ts[2:])
        print(f"{n:28s} median {ts[len(ts) // 2]:7.1f} us   min {ts[0]:7.1f} us", flush=True)


def timeline():
    """per-workgroup residency of the forward kernel (variant `timeline`)"""
    import torch
    import numpy as np
    B, N = 64, 1005
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B, N, 768, device="cuda", generator=g).bfloat16()
    res = torch.randn(B, N, 256, device="cuda", generator=g).bfloat16()
    kv = torch.full((B,), N, dtype=torch.int32, device="cuda")
    o = torch.empty(B, N, 256, device="cuda", dtype=torch.bfloat16)
    o_res = torch.empty_like(o)
    lse = torch.empty(B, 4, N, device="cuda")
    kn = torch.empty((B * N + 31) // 32, 4, device="cuda")
    P = lambda t, off=0: ctypes.c_void_p(t.data_ptr() + off)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    L = ctypes.CDLL(os.path.join(LAB, "libattn_timeline.so"))
    L.mtmp_key_norms(1, P(qkv, 512), P(kn), ctypes.c_longlong(B * N), 4, 768, st)
    for _ in range(4):
        L.mtmp_attn_fwd(1, P(qkv), P(qkv, 512), P(qkv, 1024), P(o), P(res), P(o_res), P(lse), P(kv), P(kn), B, N, 4, 768, 256,
                        ctypes.c_float(0.125), st)
    torch.cuda.synchronize()
    nwg = ((N + 255) // 256) * 4 * B
    buf = (ctypes.c_ulonglong * (4 * nwg))()
    assert L.mtmp_debug_timeline(buf, 4 * nwg) == 0
    a = np.array(buf, dtype=np.uint64).reshape(nwg, 4).astype(np.int64)
    t0 = a[:, 0].min()
    s, e = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0            # microseconds
    print(f"workgroups {nwg}: kernel span {e.max():.1f} us; residency median {np.median(e - s):.1f} us, min {(e - s).min():.1f}, max {(e - s).max():.1f}")
    print("start times: percentiles 0/25/50/75/100:", np.percentile(s, [0, 25, 50, 75, 100]).round(1))
    print("end times:   percentiles 0/25/50/75/100:", np.percentile(e, [0, 25, 50, 75, 100]).round(1))
    hw = a[:, 2]
    cu = (a[:, 3] & 0xF) * 1000 + ((hw >> 13) & 0x7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 0xF)   # xcc, se, sh, cu
    ids, cnt = np.unique(cu, return_counts=True)
    print(f"distinct (xcc, se, sh, cu) ids {len(ids)}; workgroups per id: min {cnt.min()} max {cnt.max()}; histogram {np.bincount(cnt)}")
    d = e - s
    first = s < 5.0
    buf2 = (ctypes.c_ulonglong * (4 * nwg))()
    if hasattr(L, "mtmp_debug_timeline2") and L.mtmp_debug_timeline2(buf2, 4 * nwg) == 0:
        a2 = np.array(buf2, dtype=np.uint64).reshape(nwg, 4).astype(np.int64)
        lb, le, dr = (a2[:, 0] - t0) / 100.0, (a2[:, 1] - t0) / 100.0, (a2[:, 2] - t0) / 100.0
        for nm, sel in (("round 1", first), ("round 2", ~first)):
            print(f"{nm}: prologue {np.median((lb - s)[sel]):.2f} us | key loop {np.median((le - lb)[sel]):.2f} "
                  f"(min {(le - lb)[sel].min():.2f} max {(le - lb)[sel].max():.2f}) | epilogue to last store issued "
                  f"{np.median((e - le)[sel]):.2f} | stores drained {np.median((dr - e)[sel]):.2f}")
    print("round-1 workgroups: residency by xcc:", [round(float(np.median(d[first & ((a[:, 3] & 0xF) == x)])), 1) for x in range(8)])
    print("round-2 workgroups: residency by xcc:", [round(float(np.median(d[~first & ((a[:, 3] & 0xF) == x)])), 1) for x in range(8)])
    print("round-1 residency percentiles 0/10/50/90/100:", np.percentile(d[first], [0, 10, 50, 90, 100]).round(1))
    print("round-2 residency percentiles 0/10/50/90/100:", np.percentile(d[~first], [0, 10, 50, 90, 100]).round(1))
    print("round-2 start percentiles:", np.percentile(s[~first], [0, 10, 50, 90, 100]).round(1))
    bid = np.arange(nwg)
    print("round-1 residency by blockIdx octile:", [round(float(np.median(d[first & (bid * 8 // nwg == k)])), 1) if np.any(first & (bid * 8 // nwg == k)) else None for k in range(8)])
    # the two workgroups of a CU in round 1: same duration?
    for x in range(2):
        sel = first & ((a[:, 3] & 0xF) == x)
        order = np.argsort(cu[sel])
        print(f"xcc {x} round-1 (cu id, start, dur):", [(int(c) % 1000, round(float(ss), 1), round(float(dd), 1)) for c, ss, dd in zip(cu[sel][order][:16], s[sel][order][:16], d[sel][order][:16])])
    conc = [(np.sum((s <= t) & (e > t))) for t in np.linspace(0, e.max(), 41)]
    print("resident workgroups over time:", conc)


def stamps():
    import torch
    B, N = 64, 1005
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B, N, 768, device="cuda", generator=g).bfloat16()
    res = torch.randn(B, N, 256, device="cuda", generator=g).bfloat16()
    kv = torch.full((B,), N, dtype=torch.int32, device="cuda")
    o = torch.empty(B, N, 256, device="cuda", dtype=torch.bfloat16)
    o_res = torch.empty_like(o)
    lse = torch.empty(B, 4, N, device="cuda")
    kn = torch.empty((B * N + 31) // 32, 4, device="cuda")
    P = lambda t, off=0: ctypes.c_void_p(t.data_ptr() + off)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for name in ("stamps", "stamps_occ1", "stamps_nofetch"):
        f = os.path.join(LAB, f"libattn_{name}.so")
        if not os.path.exists(f):
            continue
        L = ctypes.CDLL(f)
        L.mtmp_key_norms(1, P(qkv, 512), P(kn), ctypes.c_longlong(B * N), 4, 768, st)
        buf = (ctypes.c_ulonglong * 16)()
        for rep in range(3):
            L.mtmp_attn_fwd(1, P(qkv), P(qkv, 512), P(qkv, 1024), P(o), P(res), P(o_res), P(lse), P(kv), P(kn), B, N, 4, 768, 256,
                            ctypes.c_float(0.125), st)
            torch.cuda.synchronize()
            assert L.mtmp_debug_stamps(buf) == 0
        v = list(buf)
        n = max(v[8], 1)
        names = ["barrier", "put+fetch", "reads+slot0", "slot1", "slot2", "slot3", "slot4", "slot5"]
        print(name, "cycles per tile and wave:", {k: round(v[i] / n) for i, k in enumerate(names)}, "sum", round(sum(v[:8]) / n), "of put+fetch, put alone:", round(v[9] / n), "tiles", n)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "stamps":
        stamps()
    elif sys.argv[1] == "timeline":
        timeline()
    else:
        run(sys.argv[2] if len(sys.argv) > 2 else "fwd")
