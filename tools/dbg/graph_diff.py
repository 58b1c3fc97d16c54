import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import tests.test_gpu_parity as T
full = [[96, 96, 50, 7]] * 5
def run(g):
    ls, flat, gs = T._loop(g, 0.0, "fp32", 5, full)
    return ls, flat
le, pe = run(0)
lg, pg = run(1)
print("losses equal", le == lg)
# map flat offsets to names
from medical_tri_modal_pilot_amd.optim import FusedAdamW
args, model = T._product_model(2, 0, "fp32", hip_graph=0, dropout=0.0)
opt = FusedAdamW(model.hot_parameters(), lr=1e-4)
names = [n for n, _ in model.hot_parameters()]
for n, p, off in zip(names, opt.flat.params, opt.flat.offsets):
    a, b = pe[off:off + p.numel()], pg[off:off + p.numel()]
    if not torch.equal(a, b):
        d = (a - b).abs()
        print(f"{n:60s} ndiff {int((d > 0).sum()):6d}/{p.numel():7d} max {float(d.max()):.3e}")
