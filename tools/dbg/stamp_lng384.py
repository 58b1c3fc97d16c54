"""s_memtime stamps of the K = 384 row-panel kernel (diagnostic build -DMTMP_STAMP -> libmtmp_hip_stamp.so): prologue vs panel loop."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["MTMP_LIB"] = os.path.join(ROOT, "medical_tri_modal_pilot_amd", "libmtmp_hip_stamp.so")
import torch
from medical_tri_modal_pilot_amd import ops, _lib
L = _lib.lib()
L.mtmp_debug_stamps_lng.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 2048)()
C = 384
for rows in (6272, 12544):
    x = torch.randn(rows, C, device="cuda").bfloat16()
    lw, lb = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    for n, act in ((1152, None), (1536, "gelu")):
        w = (torch.randn(n, C, device="cuda") * 0.05).bfloat16()
        b = torch.zeros(n, device="cuda")
        f = lambda: ops.swin_ln_linear(x, lw, lb, 1e-5, w, b, act=act)
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            f()
        e1.record()
        torch.cuda.synchronize(); L.mtmp_debug_stamps_lng(buf)
        nw = (rows + 127) // 128
        pro = sorted(buf[2 * i] for i in range(nw)); loop = sorted(buf[2 * i + 1] for i in range(nw))
        print(f"rows {rows} N {n} act {act}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us/launch; prologue median {pro[nw // 2]} (max {pro[-1]}) "
              f"cycles, panel loop median {loop[nw // 2]} (min {loop[0]}, max {loop[-1]}) cycles")
