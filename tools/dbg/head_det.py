import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from medical_tri_modal_pilot_amd import ops
DEV = "cuda:0"
torch.manual_seed(0)
for B in (4, 64):
    shapes = [(256, 2), (256,), (256,), (256,), (256,), (256,), (256, 512), (256,), (256,), (256,), (1, 256), (1,)]
    base = [torch.randn(*s) * 0.1 for s in shapes]
    base[2] += 1; base[4] += 1; base[8] += 1
    cls0, age, gen = torch.randn(B, 256), torch.rand(B), torch.randint(0, 2, (B,)).float()
    w = torch.randn(B, 1)
    outs = []
    for rep in range(3):
        junk = torch.full((1 << 22,), float(rep + 1) * 1e3, device=DEV)    # perturb what recycled memory holds
        del junk
        prm = [p.clone().to(DEV).requires_grad_() for p in base]
        rm, rv = torch.zeros(256, device=DEV), torch.ones(256, device=DEV)
        c = cls0.clone().to(DEV).requires_grad_()
        out = ops.HeadFn.apply(c, age.to(DEV), gen.to(DEV), True, 0.1, 1e-5, rm, rv, *prm)
        (out * w.to(DEV)).sum().backward()
        outs.append([out.detach().clone(), c.grad.clone(), rm.clone(), rv.clone()] + [p.grad.clone() for p in prm])
    for rep in (1, 2):
        bad = [i for i, (a, b) in enumerate(zip(outs[0], outs[rep])) if not torch.equal(a, b)]
        print("B", B, "rep", rep, "tensors differing:", bad)
