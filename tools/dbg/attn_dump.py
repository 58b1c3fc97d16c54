import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from medical_tri_modal_pilot_amd import ops
out = sys.argv[1]
res = {}
for dt in (torch.float32, torch.bfloat16):
    for N, lens in ((54, None), (133, [133, 4, 5, 90]), (300, [300, 257, 64, 1]), (1005, [1005, 700])):
        g = torch.Generator().manual_seed(7 + N)
        B = 4 if lens is None else len(lens)
        qkv = torch.randn(B, N, 768, generator=g).to("cuda", dt)
        kv = None if lens is None else torch.tensor(lens, dtype=torch.int32, device="cuda")
        o, _, lse = ops.attn_fwd(qkv, kv)
        res[f"{dt}-{N}"] = (o.float().cpu(), lse.cpu())
torch.save(res, out)
