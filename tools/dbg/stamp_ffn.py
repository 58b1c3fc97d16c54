"""s_memtime stamps of the fused FFN forward (diagnostic build -DMTMP_STAMP -> libmtmp_hip_stamp.so): prologue / panel loop / epilogue."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["MTMP_LIB"] = os.path.join(ROOT, "medical_tri_modal_pilot_amd", "libmtmp_hip_stamp.so")
import torch
from medical_tri_modal_pilot_amd import ops, _lib
L = _lib.lib()
L.mtmp_debug_stamps_lng.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 2048)()
M = 64 * 1005
x = torch.randn(M, 256, device="cuda").bfloat16()
gm, bt = torch.ones(256, device="cuda"), torch.zeros(256, device="cuda")
w1, b1 = (torch.randn(1024, 256, device="cuda") * 0.05).bfloat16(), torch.zeros(1024, device="cuda")
w2, b2 = (torch.randn(256, 1024, device="cuda") * 0.03).bfloat16(), torch.zeros(256, device="cuda")
for drop in (0.0, 0.1):
    f = lambda: ops.ffn_fwd(x, gm, bt, w1, b1, w2, b2, drop_p=drop, seeds=(3, 4))
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        f()
    e1.record()
    torch.cuda.synchronize(); L.mtmp_debug_stamps_lng(buf)
    nw = min(512, (M + 127) // 128)
    col = lambda k: sorted(buf[4 * i + k] for i in range(nw))
    pro, loop, epi, t0 = col(0), col(1), col(2), col(3)
    L.mtmp_debug_stamps_tn.argtypes = [ctypes.c_void_p]
    b8 = (ctypes.c_ulonglong * 8)(); L.mtmp_debug_stamps_tn(b8)
    print("   per panel (workgroup 7, last launch): wait+barrier+DMA issue %d, S1 %d, S2 %d, S3 %d, S4 %d cycles" % tuple(b8[i] // 16 for i in range(5)))
    print(f"drop {drop}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us/launch; prologue median {pro[nw // 2]}, panel loop median {loop[nw // 2]} "
          f"(min {loop[0]}, max {loop[-1]}), epilogue median {epi[nw // 2]} cycles; start spread {t0[-1] - t0[0]} cycles "
          f"(second round starts {t0[300] - t0[0]} after the first)")
