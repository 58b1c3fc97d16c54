"""Per-step losses of the packed and the padded fusion stack at the benchmark's size (see test_packed_full_size_steps_match_padded)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, test_gpu_parity as T
g = torch.Generator().manual_seed(4)
lens = [torch.randint(3, 1001, (64,), generator=g).tolist() for _ in range(3)]
lens[1][0] = 1000
for mode in (0, 1):
    ls, ps, gs = T._loop(1, 0.0, "bf16", 3, lens, L=6, B=64, T=1000, batch_size=64, pack_rows=mode, skip_missing_images=mode)
    print(mode, ls)
for mode in ((1, 0), (0, 1)):
    ls, ps, gs = T._loop(1, 0.0, "bf16", 3, lens, L=6, B=64, T=1000, batch_size=64, pack_rows=mode[0], skip_missing_images=mode[1])
    print(mode, ls)
