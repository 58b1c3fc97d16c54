"""Where does a workgroup of the LDS-DMA weight-gradient kernel spend its time?  Diagnostic build with s_memtime stamps
(-DMTMP_STAMP -> libmtmp_hip_stamp.so); wave 0 of group 0 of every workgroup."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["MTMP_LIB"] = os.path.join(ROOT, "medical_tri_modal_pilot_amd", "libmtmp_hip_stamp.so")
import torch
from medical_tri_modal_pilot_amd import ops, _lib
M = 64 * 1005
L = _lib.lib()
L.mtmp_debug_stamps_lng.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 2048)()
for n, k in ((768, 256), (256, 1024)):
    dy, x = torch.randn(M, n, device="cuda").bfloat16(), torch.randn(M, k, device="cuda").bfloat16()
    f = lambda: ops.gemm_tn(dy, x)
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        f()
    e1.record()
    torch.cuda.synchronize(); L.mtmp_debug_stamps_lng(buf)
    nw = 128
    names = ["prologue", "own wait (vmcnt | lgkmcnt)", "work (DMA issue | fragments + MFMAs)", "realtime ticks (100 MHz)", "epilogue", "total", "barrier", "stages"]
    print(f"gemm_tn[{n},{k}]: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us/launch (incl. reduce)")
    for g in (0, 1):
        print(f"  {'matrix' if g == 0 else 'loader'} wave 0:")
        for j, nm in enumerate(names):
            v = sorted(buf[16 * i + 8 * g + j] for i in range(nw))
            print(f"   {nm:40s} median {v[nw // 2]:7d}  min {v[0]:7d}  max {v[-1]:7d}")
        tot = sorted(buf[16 * i + 8 * g + 5] / max(1, buf[16 * i + 8 * g + 3]) * 100e6 / 1e9 for i in range(nw))
        print(f"   in-kernel clock (s_memtime / s_memrealtime)  median {tot[nw // 2]:.2f} GHz")
