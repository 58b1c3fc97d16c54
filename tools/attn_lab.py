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

    work = torch.zeros(16, dtype=torch.int32, device="cuda")
    VP = ctypes.c_void_p

    def fwd(L, name=""):
        if hasattr(L, "mtmp_attn_work_words") and not name.endswith("static"):       # persistent kernel with its work queue
            arr = lambda t, off=0: (VP * 1)(t.data_ptr() + off)
            ints = lambda v: (ctypes.c_int * 1)(v)
            rc = L.mtmp_attn_fwd_grouped(1, 1, arr(qkv), arr(qkv, 512), arr(qkv, 1024), arr(o), arr(res), arr(o_res), arr(lse), arr(kv),
                                         None, arr(kn), ints(N), ints(768), ints(256), B, 4, ctypes.c_float(0.125), P(work), st)
        else:
            rc = L.mtmp_attn_fwd(1, P(qkv), P(qkv, 512), P(qkv, 1024), P(o), P(res), P(o_res), P(lse), P(kv), P(kn), B, N, 4, 768, 256,
                                 ctypes.c_float(0.125), st)
        assert rc == 0

    def bwd(L):
        rc = L.mtmp_attn_bwd(1, P(qkv), P(qkv, 512), P(qkv, 1024), P(o), P(do), P(lse), P(kv), P(dqkv), P(dqkv, 512), P(dqkv, 1024),
                             P(delta), B, N, 4, 768, 256, 256, 768, ctypes.c_float(0.125), st)
        assert rc == 0
    fn = (lambda L, n: fwd(L, n)) if what == "fwd" else (lambda L, n: bwd(L))
    fwd(first)
    torch.cuda.synchronize()
    ref_o = None
    if what == "fwd":                      # outputs of every variant against the first one (ablation builds differ by design)
        for n, L in libs.items():
            o.zero_(); o_res.zero_(); lse.zero_()
            fwd(L, n)
            torch.cuda.synchronize()
            cur = (o.float().clone(), o_res.float().clone(), lse.clone())
            if ref_o is None:
                ref_o = cur
            else:
                print(f"{n:28s} max |diff| vs {next(iter(libs))}: o {float((cur[0] - ref_o[0]).abs().max()):.3g} o_res "
                      f"{float((cur[1] - ref_o[1]).abs().max()):.3g} lse {float((cur[2] - ref_o[2]).abs().max()):.3g}; work words {work.tolist()[:9]}")
    times = {n: [] for n in libs}
    for rnd in range(12):
        for n, L in libs.items():
            fn(L, n)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn(L, n)
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


def clock(name="clock"):
    """in-kernel clock (s_memtime / s_memrealtime) and phase durations of the forward, variant `clock` (or `clock_*`)"""
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
    L = ctypes.CDLL(os.path.join(LAB, f"libattn_{name}.so"))
    L.mtmp_key_norms(1, P(qkv, 512), P(kn), ctypes.c_longlong(B * N), 4, 768, st)
    for _ in range(400):                    # the clock settles under sustained load (MI355X_MICROARCH, DVFS give-back item 6)
        L.mtmp_attn_fwd(1, P(qkv), P(qkv, 512), P(qkv, 1024), P(o), P(res), P(o_res), P(lse), P(kv), P(kn), B, N, 4, 768, 256,
                        ctypes.c_float(0.125), st)
    torch.cuda.synchronize()
    nwg = ((N + 255) // 256) * 4 * B
    buf = (ctypes.c_ulonglong * (8 * nwg))()
    assert L.mtmp_debug_read(buf, 8 * nwg) == 0
    a = np.array(buf, dtype=np.uint64).reshape(nwg, 4, 2).astype(np.int64)
    cyc, real = a[:, :, 0], a[:, :, 1]
    t0 = real[:, 0].min()
    for i, nm in enumerate(("prologue", "key loop", "epilogue")):
        dc, dr = cyc[:, i + 1] - cyc[:, i], (real[:, i + 1] - real[:, i]) / 100.0
        ok = dr > 0
        print(f"{nm:9s}: median {np.median(dr):7.2f} us (min {dr.min():.2f} max {dr.max():.2f}); {np.median(dc):9.0f} cycles; "
              f"clock {np.median(dc[ok] / dr[ok]) / 1e3:.3f} GHz")
    s_, e_ = (real[:, 0] - t0) / 100.0, (real[:, 3] - t0) / 100.0
    print(f"kernel span {e_.max():.1f} us; workgroup residency median {np.median(e_ - s_):.1f}; starts 0/50/100 % {np.percentile(s_, [0, 50, 100]).round(1)}; "
          f"ends {np.percentile(e_, [0, 50, 100]).round(1)}")
    late = s_ > 5.0
    for nm, sel in (("first round", ~late), ("second round", late)):
        if sel.any():
            dr = (real[sel, 2] - real[sel, 1]) / 100.0
            print(f"  {nm}: {sel.sum()} workgroups, key loop median {np.median(dr):.2f} us, residency {np.median((e_ - s_)[sel]):.2f} us")


def pclock(name="pclock"):
    """per-item real-time stamps of the persistent forward (variants pclock / pclock_static)"""
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
    work = torch.zeros(16, dtype=torch.int32, device="cuda")
    P = lambda t, off=0: ctypes.c_void_p(t.data_ptr() + off)
    VP = ctypes.c_void_p
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    L = ctypes.CDLL(os.path.join(LAB, f"libattn_{name}.so"))
    L.mtmp_key_norms(1, P(qkv, 512), P(kn), ctypes.c_longlong(B * N), 4, 768, st)
    arr = lambda t, off=0: (VP * 1)(t.data_ptr() + off)
    ints = lambda v: (ctypes.c_int * 1)(v)

    def fwd():
        rc = L.mtmp_attn_fwd_grouped(1, 1, arr(qkv), arr(qkv, 512), arr(qkv, 1024), arr(o), arr(res), arr(o_res), arr(lse), arr(kv), None,
                                     arr(kn), ints(N), ints(768), ints(256), B, 4, ctypes.c_float(0.125),
                                     None if name.endswith("static") else P(work), st)
        assert rc == 0
    L.mtmp_debug_clear()
    for _ in range(200):                     # back to back: the stamps of the LAST launch stay (warm clocks)
        fwd()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (32 * 1024))()
    assert L.mtmp_debug_read(buf, 32 * 1024) == 0
    a = np.array(buf, dtype=np.uint64).reshape(1024, 4, 8).astype(np.float64) / 100.0       # [workgroup][item][stamp] in us
    live = a[:, 0, 0] > 0
    t0 = a[live, 0, 0].min()
    a = np.where(a > 0, a - t0, np.nan)
    print(f"{name}: workgroups that ran: {int(live.sum())}; items per workgroup histogram {np.bincount((~np.isnan(a[live, :, 0])).sum(1))}")
    for k in range(4):
        sel = ~np.isnan(a[:, k, 3])
        if not sel.any():
            continue
        x = a[sel, k]
        print(f"item {k}: n {int(sel.sum())}: start median {np.median(x[:, 0]):6.1f} | first phase (to loop start) {np.median(x[:, 1] - x[:, 0]):5.2f} | key loop "
              f"{np.median(x[:, 2] - x[:, 1]):5.2f} (min {np.min(x[:, 2] - x[:, 1]):.2f} max {np.max(x[:, 2] - x[:, 1]):.2f}) | epilogue {np.median(x[:, 3] - x[:, 2]):5.2f} | end median {np.median(x[:, 3]):6.1f} max {np.max(x[:, 3]):6.1f}")
        print(f"        loop start -> all tiles but the last {np.nanmedian(x[:, 4] - x[:, 1]):5.2f} | publish + last tile {np.nanmedian(x[:, 5] - x[:, 4]):5.2f} | decode + request next "
              f"{np.nanmedian(x[:, 6] - x[:, 5]):5.2f} | drain {np.nanmedian(x[:, 2] - x[:, 6]):5.2f}")


def clock_bwd(name="clock_bwd"):
    """in-kernel clock and phase durations of the two backward kernels (variant `clock_bwd`)"""
    import torch
    import numpy as np
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
    L = ctypes.CDLL(os.path.join(LAB, f"libattn_{name}.so"))
    L.mtmp_key_norms(1, P(qkv, 512), P(kn), ctypes.c_longlong(B * N), 4, 768, st)
    L.mtmp_attn_fwd(1, P(qkv), P(qkv, 512), P(qkv, 1024), P(o), P(res), P(o_res), P(lse), P(kv), P(kn), B, N, 4, 768, 256, ctypes.c_float(0.125), st)
    for _ in range(150):
        L.mtmp_attn_bwd(1, P(qkv), P(qkv, 512), P(qkv, 1024), P(o), P(do), P(lse), P(kv), P(dqkv), P(dqkv, 512), P(dqkv, 1024),
                        P(delta), B, N, 4, 768, 256, 256, 768, ctypes.c_float(0.125), st)
    torch.cuda.synchronize()
    nwg = ((N + 127) // 128) * 4 * B
    for which, kname in ((0, "dQ"), (1, "dK/dV")):
        buf = (ctypes.c_ulonglong * (8 * nwg))()
        assert L.mtmp_debug_read(buf, 8 * nwg, which) == 0
        a = np.array(buf, dtype=np.uint64).reshape(nwg, 4, 2).astype(np.int64)
        cyc, real = a[:, :, 0], a[:, :, 1]
        t0 = real[:, 0].min()
        print(f"== {kname}: {nwg} workgroups")
        for i, nm in enumerate(("prologue", "tile loop", "epilogue")):
            dc, dr = cyc[:, i + 1] - cyc[:, i], (real[:, i + 1] - real[:, i]) / 100.0
            ok = dr > 0
            print(f"{nm:9s}: median {np.median(dr):7.2f} us (min {dr.min():.2f} max {dr.max():.2f}); {np.median(dc):9.0f} cycles; clock {np.median(dc[ok] / dr[ok]) / 1e3:.3f} GHz")
        s_, e_ = (real[:, 0] - t0) / 100.0, (real[:, 3] - t0) / 100.0
        print(f"kernel span {e_.max():.1f} us; residency median {np.median(e_ - s_):.1f}; starts 0/25/50/75/100 % {np.percentile(s_, [0, 25, 50, 75, 100]).round(1)}; ends {np.percentile(e_, [0, 25, 50, 75, 100]).round(1)}")


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
    elif sys.argv[1] == "pclock":
        pclock(*sys.argv[2:3])
    elif sys.argv[1] == "clock_bwd":
        clock_bwd(*sys.argv[2:3])
    elif sys.argv[1] == "clock":
        clock(*sys.argv[2:3])
    elif sys.argv[1] == "stamps":
        stamps()
    elif sys.argv[1] == "timeline":
        timeline()
    else:
        run(sys.argv[2] if len(sys.argv) > 2 else "fwd")
