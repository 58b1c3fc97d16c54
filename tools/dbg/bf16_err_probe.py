"""bf16 build against fp32 build (both HIP), per-tensor relative L2 error and cosine of the parameter gradients of one train step,
for a sweep of (B, T, L) and of the bf16 path's layout switches: where does the error of the benchmarked build come from?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch
import test_gpu_parity as TG
from medical_tri_modal_pilot_amd import synthetic, tuning

cases = [tuple(int(v) for v in c.split(",")) for c in sys.argv[1:]] or [(4, 32, 2), (64, 64, 2), (64, 1000, 2), (64, 64, 6), (64, 1000, 6)]


def stats(gb, gf):
    med = float(np.median([float(v.norm()) for v in gf.values()]))
    keys = [n for n, g in gf.items() if float(g.norm()) >= 1e-4 * med]
    l2 = np.array([float((gb[n] - gf[n]).norm() / gf[n].norm()) for n in keys])
    cos = np.array([float((gb[n] * gf[n]).sum() / (gb[n].norm() * gf[n].norm())) for n in keys])
    w = int(l2.argmax())
    return (f"L2 rel median {np.median(l2):.3f} p95 {np.percentile(l2, 95):.3f} worst {l2.max():.3f} ({keys[w]}) | "
            f"cos median {np.median(cos):.4f} p5 {np.percentile(cos, 5):.4f} worst {cos.min():.4f}")


for B, T, L in cases:
    bt = synthetic.make_batch(1234, B, T, ragged=True, missing_mode="mixed")
    lf, gf, _ = TG._one_train_step("fp32", B, T, L, batch=bt)
    lb, gb, _ = TG._one_train_step("bf16", B, T, L, batch=bt)
    print(f"B={B} T={T} L={L}: loss diff {abs(lb - lf):.2e} | {stats(gb, gf)}", flush=True)
    if (B, T, L) == cases[-1]:
        over = dict(pack_rows=0, skip_missing_images=0)
        orig = TG._product_model

        def patched(L_, multi, dtype, **kw):
            if dtype == "bf16":
                kw.update(over)
            return orig(L_, multi, dtype, **kw)
        TG._product_model = patched
        tuning.GROUPED_LAUNCHES = False
        lb2, gb2, _ = TG._one_train_step("bf16", B, T, L, batch=bt)
        print(f"   padded rows, all images encoded, one launch per stream: loss diff {abs(lb2 - lf):.2e} | {stats(gb2, gf)}")
        print(f"   that variant against the default bf16 path: {stats(gb2, gb)}")
        TG._product_model = orig
        tuning.GROUPED_LAUNCHES = True
