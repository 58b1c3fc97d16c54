"""Synthetic tri-modal batches and closed-form deterministic weights.

``make_batch`` is the seeded recipe of §8d (what ``--synthetic 1`` trains on and what bench.py
measures); ``fill_tensor`` / ``fill_state_dict`` overwrite every float tensor of a state_dict with
``scale * hash(name, i)`` (exact integer hash -> uniform [-1,1)), so that the golden generator (build
container, real reference), the tests (CPU restatement, HIP path) and the benchmark can all rebuild the exact
same 42 M parameters from nothing -- weights are never committed.  tests/golden/filler.py re-exports
this module.
"""
import zlib

import numpy as np
import torch


def _hash_uniform(name: str, n: int) -> np.ndarray:
    """Exact integer hash (murmur3 finaliser) of (crc32(name), i) -> float64 in [-1, 1)."""
    seed = np.uint64(zlib.crc32(name.encode()))
    M = np.uint64(0xFFFFFFFF)
    x = (np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B1) + seed * np.uint64(0x85EBCA6B)) & M
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & M
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & M
    x ^= x >> np.uint64(16)
    return x.astype(np.float64) / 2147483648.0 - 1.0


def fill_tensor(name: str, t: torch.Tensor) -> torch.Tensor:
    """Returns a new fp32 tensor of t's shape (integer tensors are returned unchanged)."""
    if not t.is_floating_point():
        return t.clone()
    n = t.numel()
    w = _hash_uniform(name, n)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "running_var":
        w = 1.0 + 0.25 * w
    elif leaf == "running_mean":
        w = 0.1 * w
    elif leaf in ("gamma",) or (leaf == "weight" and t.dim() == 1):
        w = 1.0 + 0.1 * w                       # LayerNorm / BatchNorm scales
    elif leaf in ("bias", "beta"):
        w = 0.05 * w
    elif "relative_position_bias_table" in name:
        w = 0.2 * w
    elif "cls_token" in name or name.endswith("bottlenecks") or name.endswith("ie_feat.weight"):
        w = 0.7 * w
    elif name.endswith(".pe"):
        return t.clone()                        # sinusoid buffer: keep as built
    elif t.dim() >= 2:
        fan_in = int(np.prod(t.shape[1:]))
        w = w * (1.7 / np.sqrt(max(fan_in, 1)))
    else:
        w = 0.1 * w
    return torch.from_numpy(w.astype(np.float32)).reshape(t.shape)


def fill_state_dict(sd):
    return {k: fill_tensor(k, v) for k, v in sd.items()}


def make_batch(seed: int, B: int, T: int, *, ragged: bool = True, missing_mode: str = "mixed",
               multiimages: int = 0, txt_tokens: int = 128, img_size: int = 224, n_images: int = 3):
    """Synthetic tri-modal batch following SURVEY.md §8d (seeded, CPU generator).
    Returns a dict of CPU tensors with the trainer-level (post-unpack) meaning."""
    g = torch.Generator().manual_seed(seed)
    U = lambda *s: torch.rand(*s, generator=g)
    if ragged:
        lens = torch.randint(3, T + 1, (B,), generator=g)
        lens[0] = T
    else:
        lens = torch.full((B,), T, dtype=torch.long)
    x = torch.zeros(B, T, 3)
    for b in range(B):
        n = int(lens[b])
        x[b, :n, 0] = torch.sort(-24.0 * U(n))[0]
        x[b, :n, 1] = U(n)
        x[b, :n, 2] = torch.randint(0, 18, (n,), generator=g).float()
    x = x.half().float()                                     # 2_train.py:164 rounding
    age, gen = U(B), torch.randint(0, 2, (B,), generator=g).float()
    if missing_mode == "mixed":
        mnum = torch.multinomial(torch.tensor([0.4, 0.2, 0.2, 0.2]), B, True, generator=g)
        mnum[: min(4, B)] = torch.arange(min(4, B))          # make sure every pattern appears
    elif missing_mode == "none":
        mnum = torch.zeros(B, dtype=torch.long)
    else:
        mnum = torch.full((B,), int(missing_mode), dtype=torch.long)
    txt_missing = (mnum == 1) | (mnum == 3)
    img_missing = (mnum == 2) | (mnum == 3)
    txt_len = torch.randint(1, txt_tokens - 1, (B,), generator=g)
    txt_len[txt_missing] = 0
    txt = torch.randn(B, txt_tokens, 768, generator=g)
    txt = txt * (torch.arange(txt_tokens).view(1, -1, 1) < txt_len.view(-1, 1, 1))
    K = int(n_images) if multiimages else 1     # the reference hard-codes 3 (tri_mbt_vsltcls.py:161-162)
    img = U(B, K, 1, img_size, img_size)
    img_time = (-10.0 * U(B, K)).half().float()
    if multiimages:
        absent = U(B, K) < 0.3
        absent[:, 0] = False
        img_time[absent] = 10.0
        img = img * (~absent).view(B, K, 1, 1, 1)
    img[img_missing] = 0
    img_time[img_missing] = 10.0 if multiimages else -1.0
    if not multiimages:
        img, img_time = img[:, 0], img_time[:, 0]
    txt_time = -torch.randint(3, 101, (B,), generator=g).float()
    y = torch.randint(0, 2, (B,), generator=g)
    missing = torch.stack([torch.zeros(B), img_missing.float(), txt_missing.float()], 1)
    return dict(x=x, age=age, gen=gen, input_lengths=lens, txt=txt, txt_lengths=txt_len, img=img,
                img_time=img_time, txt_time=txt_time, y=y, missing=missing, missing_num=mnum)
