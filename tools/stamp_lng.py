"""Where does a wave of the LN-fused GEMM (bf16, LDS-DMA build) spend its time?  Diagnostic build with s_memtime stamps
(-DMTMP_STAMP -> libmtmp_hip_stamp.so); s_memtime ticks are shader cycles."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MTMP_LIB"] = os.path.join(ROOT, "medical_tri_modal_pilot_amd", "libmtmp_hip_stamp.so")
import torch
from medical_tri_modal_pilot_amd import ops, _lib
M = 64 * 1005
L = _lib.lib()
L.mtmp_debug_stamps_lng.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 2048)()
x = torch.randn(M, 256, device="cuda").bfloat16()
gm, bt = torch.ones(256, device="cuda"), torch.zeros(256, device="cuda")
for n, relu, drop in ((768, False, 0.0), (1024, True, 0.0), (1024, True, 0.1)):
    w = (torch.randn(n, 256, device="cuda") * 0.05).bfloat16()
    b = torch.zeros(n, device="cuda")
    f = lambda: ops.ln_gemm(x, gm, bt, w, b, n, relu=relu, drop_p=drop, seed=5)
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        f()
    e1.record()
    torch.cuda.synchronize(); L.mtmp_debug_stamps_lng(buf)
    nw = (M + 127) // 128
    pro = sorted(buf[2 * i] for i in range(nw)); loop = sorted(buf[2 * i + 1] for i in range(nw))
    print(f"ln_gemm[N={n}, relu={relu}, drop={drop}]: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us/launch; per wave: prologue "
          f"median {pro[nw // 2]} (max {pro[-1]}) cycles, panel loop median {loop[nw // 2]} (min {loop[0]}, max {loop[-1]}) cycles")
