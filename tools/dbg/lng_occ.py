"""ln_gemm (FFN1 shape, relu + dropout) timed at 128 / 256 / 503 workgroups: does the second workgroup on a CU overlap?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from medical_tri_modal_pilot_amd import ops
from tools.bench_kernels import timeit
BF = torch.bfloat16
for wgs in (128, 256, 384, 503, 512, 768, 1024):
    M = wgs * 128
    x = torch.randn(M, 256, device="cuda").to(BF)
    w = (torch.randn(1024, 256, device="cuda") * 0.05).to(BF)
    b = torch.zeros(1024, device="cuda")
    gm, bt = torch.ones(256, device="cuda"), torch.zeros(256, device="cuda")
    for drop in (0.0, 0.1):
        t = timeit(lambda: ops.ln_gemm(x, gm, bt, w, b, 1024, relu=True, drop_p=drop, seed=3), 7)
        print(f"wgs {wgs:5d} drop {drop}: {t*1e3:7.1f} us  ({2.0*M*1024*256/t/1e9:6.1f} TF/s)", flush=True)
