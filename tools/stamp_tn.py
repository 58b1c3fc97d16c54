"""Where does a 64-token step of the weight-gradient GEMM (gemm_tn) spend its cycles?  Diagnostic build
(make -C medical_tri_modal_pilot_amd/csrc stamp) with s_memtime stamps; s_memtime ticks at 100 MHz."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MTMP_LIB"] = os.path.join(ROOT, "medical_tri_modal_pilot_amd", "libmtmp_hip_stamp.so")
import torch
from medical_tri_modal_pilot_amd import ops, _lib
M = 64 * 1005
L = _lib.lib()
L.mtmp_debug_stamps_tn.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 8)()
for n, k in ((768, 256), (256, 1024)):
    dy = torch.randn(M, n, device="cuda").bfloat16()
    x = torch.randn(M, k, device="cuda").bfloat16()
    for _ in range(3):
        ops.gemm_tn(dy, x)
    torch.cuda.synchronize(); L.mtmp_debug_stamps_tn(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.gemm_tn(dy, x)
    e1.record()
    torch.cuda.synchronize(); L.mtmp_debug_stamps_tn(buf)
    steps = buf[4]
    names = ["barrier 1", "load wait + transpose + ds_write", "barrier 2", "fetch issue + MFMA"]
    tot = sum(buf[i] for i in range(4))
    print(f"gemm_tn[{n},{k},M]: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us/launch (incl. reduce), steps/wave {steps / 5:.0f}")
    for i, nm in enumerate(names):
        print(f"  {nm:34s} {buf[i] / steps:9.1f} ticks/step/wave  {100 * buf[i] / tot:5.1f}%")
    print("  total ticks per step per wave", tot / steps, "(s_memtime: 100 MHz -> 10 ns per tick)")
