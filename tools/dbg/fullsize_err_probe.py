"""Per-tensor gradient error of the fp32 HIP train step against the oracle for a few (B, T, L): which tensors leave the 1e-4 band,
and from which batch size on (round 5: the text stream's layer-0 tensors at B = 64)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch
import test_gpu_parity as TG
from oracle import tri_mbt_oracle as O
from medical_tri_modal_pilot_amd import synthetic

cases = [tuple(int(v) for v in c.split(",")) for c in sys.argv[1:]] or [(16, 64, 2), (64, 64, 2)]
for B, T, L in cases:
    for mode in ("mixed", "none"):
        bt = synthetic.make_batch(1234, B, T, ragged=True, missing_mode=mode)
        loss, grads, _ = TG._one_train_step("fp32", B, T, L, batch=bt)
        tr = O.OracleTrainer(TG._model_sd(L), O.Cfg(n_layers=L), lr_init=1e-5, batch_size=B, iters_per_epoch=10)
        ref = tr.step(bt, 1)
        errs = TG._tensor_errors(grads, tr.grads)
        print(f"B={B} T={T} L={L} missing={mode}: loss diff {abs(loss - ref):.2e} median {errs[len(errs)//2][0]:.2e} over 1e-3: {sum(1 for e,_ in errs if e > 1e-3)}")
        for e, n in errs[:6]:
            g, r = grads[n].float(), tr.grads[n].float()
            d = (g - r).abs()
            print(f"   {e:.2e} {n}  |ref| {float(r.norm()):.3e}  max|d| {float(d.max()):.3e} at {int(d.argmax())} of {d.numel()}  frac>1e-3*max|ref|: {float((d > 1e-3 * r.abs().max()).float().mean()):.4f}")
