import torch, sys
sys.path.insert(0, '.')
from medical_tri_modal_pilot_amd import ops
DEV="cuda"; M=int(sys.argv[1]) if len(sys.argv) > 1 else 300
g = torch.Generator(device=DEV).manual_seed(M); bf=torch.bfloat16
x = (torch.randn(M, 256, generator=g, device=DEV) * 2 + 0.3).to(bf)
gam, bet = 1 + 0.1 * torch.randn(256, generator=g, device=DEV), 0.1 * torch.randn(256, generator=g, device=DEV)
w1, b1 = (torch.randn(1024, 256, generator=g, device=DEV) / 16).to(bf), 0.1 * torch.randn(1024, generator=g, device=DEV)
w2, b2 = (torch.randn(256, 1024, generator=g, device=DEV) / 32).to(bf), 0.1 * torch.randn(256, generator=g, device=DEV)
for p in (0.0, 0.1):
    h0, xn0, st0, sg0 = ops.ln_gemm(x, gam, bet, w1, b1, 1024, relu=True, drop_p=p, seed=11, want_signs=True)
    out0 = ops.gemm_nt(h0, w2, b2, res2d=x, drop_p=p, seed=12)
    out1, h1, xn1, st1, sg1 = ops.ffn_fwd(x, gam, bet, w1, b1, w2, b2, drop_p=p, seeds=(11, 12))
    print("p", p, "stats maxdiff", (st1-st0).abs().max().item(), "rows differing", ((st1!=st0).any(1)).sum().item(),
          "h equal", torch.equal(h1,h0), "h diff frac", (h1!=h0).float().mean().item(), "signs eq", torch.equal(sg1,sg0),
          "out maxdiff", (out1.float()-out0.float()).abs().max().item(), "out diff frac", (out1!=out0).float().mean().item())
    print(st1[:3], st0[:3])
    d = (st1 != st0)
    print("  mean differs in", d[:, 0].sum().item(), "rows, rstd in", d[:, 1].sum().item(), "; xn equal", torch.equal(xn1, xn0),
          "xn diff frac", (xn1 != xn0).float().mean().item())
