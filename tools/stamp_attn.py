"""Where does a K/V-tile iteration of the attention forward spend its cycles?  Uses the diagnostic
build (make -C medical_tri_modal_pilot_amd/csrc stamp) whose kernel carries s_memtime stamps."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MTMP_LIB"] = os.path.join(ROOT, "medical_tri_modal_pilot_amd", "libmtmp_hip_stamp.so")
import torch
from medical_tri_modal_pilot_amd import ops, _lib
B, N = 64, 1005
qkv = torch.randn(B, N, 768, device="cuda").bfloat16()
kv = torch.full((B,), N, dtype=torch.int32, device="cuda")
L = _lib.lib()
L.mtmp_debug_stamps.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 8)()
for _ in range(3):
    ops.attn_fwd(qkv, kv)
torch.cuda.synchronize(); L.mtmp_debug_stamps(buf)
for _ in range(5):
    ops.attn_fwd(qkv, kv)
torch.cuda.synchronize(); L.mtmp_debug_stamps(buf)
tiles = buf[4]
names = ["barrier+LDS puts", "fetch issue + S=K.Q^T", "softmax", "P.V"]
tot = sum(buf[i] for i in range(4))
for i, n in enumerate(names):
    print(f"{n:24s} {buf[i]/tiles:9.1f} cycles/tile/wave  {100*buf[i]/tot:5.1f}%")
print("total per tile per wave", tot / tiles)
