"""In-kernel s_memtime stamps of attn_bwd_dkdv64_kernel (lab build libmtmp_ab_CLOCK.so): per-slot cycles of the first workgroups."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["MTMP_LIB"] = os.path.join(ROOT, "medical_tri_modal_pilot_amd", "libmtmp_ab_CLOCK.so")
import numpy as np
import torch
from medical_tri_modal_pilot_amd import ops, _lib
B, N = 64, 1005
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, N, 768, device="cuda", generator=g).bfloat16()
do = torch.randn(B, N, 256, device="cuda", generator=g).bfloat16()
kv = torch.full((B,), N, dtype=torch.int32, device="cuda")
o, _, lse = ops.attn_fwd(qkv, kv, knorm=ops.key_norms(qkv))
for _ in range(20):
    ops.attn_bwd(qkv, o, do, lse, kv)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["MTMP_LIB"])
n = 64 * 4 * 160
buf = (ctypes.c_longlong * n)()
assert lib.mtmp_dbg_read_stamps(buf, n) == 0
s = np.frombuffer(buf, dtype=np.int64).reshape(64, 4, 160)
life = s[:, :, 151] - s[:, :, 0]
print("wave lifetime cycles: median", np.median(life), "min", life.min(), "max", life.max())
print("prologue (0->1):", np.median(s[:, :, 1] - s[:, :, 0]), " epilogue (150->151):", np.median(s[:, :, 151] - s[:, :, 150]))
names = ["stage(2->3)", "slotA", "slotB", "slotC", "barrier(6->7)", "slotD", "->next tile"]
for it in (0, 1, 5, 10, 14):
    b = 2 + 8 * it
    d = [np.median(s[:, :, b + k + 1] - s[:, :, b + k]) for k in range(6)] + [np.median(s[:, :, b + 8] - s[:, :, b + 6 + 0 + 0]) if it < 15 else 0]
    print(f"tile {it}: " + "  ".join(f"{n_} {int(v)}" for n_, v in zip(names, d)), " tile total", int(np.median(s[:, :, b + 8] - s[:, :, b])) if it < 15 else "")
for wv in range(4):
    b = 2 + 8 * 5
    print(f"wave {wv} tile 5:", [int(np.median(s[:, wv, b + k + 1] - s[:, wv, b + k])) for k in range(6)])
