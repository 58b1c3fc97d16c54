"""Swin-T chest-X-ray encoder.  The tri-modal model runs it frozen under no_grad (the forward-only kernels below); the sibling
models that TRAIN it (bi_vsltimg_mbt_v1.py:203-206, tri_mbt_v2.py:208-211 call it with gradients) take `forward_train`: the
same blocks as autograd nodes over libmtmp_hip.so kernels (ops.LinearFn, LayerNormRowsFn, GeluFn, WindowAttnFn: mtmp_gemm_nt /
mtmp_gemm_tn, mtmp_layernorm_rows(_bwd), mtmp_gelu_fwd / _bwd, mtmp_swin_window_attn(_bwd)); residual adds, StochasticDepth
scaling and the 2x2 patch gather are torch glue.

Parameter tree / state_dict keys equal the reference's torchvision fork
(builder/models/src/swin_transformer.py:503-654: 1-channel 4x4/4 stem, depths [2,2,6,2],
heads [3,6,12,24], window 7, returns the normalised [B,7,7,768] map -- no pooling/head).

Every block runs on libmtmp_hip.so kernels over ONE un-shifted NHWC map:
    stem            mtmp_swin_stem_fwd      conv 4x4/4 as implicit GEMM + LayerNorm(96)
    norm1 / norm2   mtmp_layernorm_rows     (patch merging: the 2x2 gather is fused into it)
    qkv / proj /    mtmp_gemm_nt            bias, exact GELU, StochasticDepth row scale and the
    mlp / merge                             residual add in the epilogue
    W-MSA / SW-MSA  mtmp_swin_window_attn   shift + window partition + bias/mask + softmax + PV +
                                            reverse as address arithmetic (no roll/permute copies)
The additive table (relative-position bias + shift mask) is constant per block: built once on
the host and cached.  Feature maps whose side is not a multiple of 7 (--image-size 512: 128 / 64 /
32 / 16 tokens a side) take the reference's zero-padded windows (:150-152): a padded token is a zero
vector behind norm1, so its q / k / v are the projection's bias -- the qkv map is laid into a
bias-filled map of the padded size (two torch copies per block), the window kernel runs on that map
and the result is cropped.  Odd-sized maps are zero-padded in front of a patch merging (:34-44).
The present-images-only form (`slots`) needs maps that are multiples of the window at every stage.
"""
import contextlib
from typing import List

import torch
import torch.nn as nn

from medical_tri_modal_pilot_amd import ops

WS = 7
PAD_LOGIT = -30000.0

# Both on in the product; tools/dbg A/B scripts flip these module attributes (no environment switches in the package).
_SPLIT_TAIL = True     # stages 3-4 as two half batches on two streams
_FUSED_MLP = True      # mtmp_swin_ln_linear / mtmp_swin_mlp (C = 96 / 192)
_FUSED_ATTN = True     # mtmp_swin_attn_block: norm1 -> qkv -> window attention -> proj -> residual in one launch


def _w(p: torch.Tensor, dtype) -> torch.Tensor:
    """Frozen-encoder weights in the compute dtype, converted once per (tensor, version).  The copy lives ON the
    parameter object (a module-level dict keyed by id(p) outlived its parameters: a later module could get a recycled
    id + data_ptr and be handed another layer's weights)."""
    if p.dtype == dtype:
        return p
    cache = getattr(p, "_mtmp_cast", None)
    if cache is None:
        cache = p._mtmp_cast = {}
    hit = cache.get(dtype)
    if hit is None or hit[0] != (p._version, p.data_ptr(), tuple(p.shape)):
        hit = ((p._version, p.data_ptr(), tuple(p.shape)), p.detach().to(dtype).contiguous())
        cache[dtype] = hit
    return hit[1]


def _relative_position_index(ws: int) -> torch.Tensor:
    ax = torch.arange(ws)
    grid = torch.stack(torch.meshgrid(ax, ax, indexing="ij")).flatten(1)          # [2, ws*ws]
    rel = grid[:, :, None] - grid[:, None, :]
    return ((rel[0] + ws - 1) * (2 * ws - 1) + (rel[1] + ws - 1)).flatten()


def _shift_mask(Hp: int, Wp: int, ws: int, sh: int, sw: int) -> torch.Tensor:
    """[nW, ws*ws, ws*ws] additive mask (0 / -100) of the shifted windows (swin_transformer.py:190-203)."""
    region = torch.zeros(Hp, Wp)
    k = 0
    for hs in ((0, -ws), (-ws, -sh), (-sh, None)):
        for ws_ in ((0, -ws), (-ws, -sw), (-sw, None)):
            region[hs[0]:hs[1], ws_[0]:ws_[1]] = k
            k += 1
    region = region.view(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    diff = region[:, None, :] - region[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


class Permute(nn.Module):
    def __init__(self, dims: List[int]):
        super().__init__()
        self.dims = dims

    def forward(self, x):
        return x.permute(self.dims)


class StochasticDepth(nn.Module):
    """Row-mode stochastic depth (torchvision.ops.StochasticDepth semantics), active in train mode only:
    `row_scale(n)` is the per-sample factor applied in the projection GEMM's epilogue."""

    def __init__(self, p: float, mode: str = "row"):
        super().__init__()
        self.p, self.mode = p, mode

    def row_scale(self, n: int, device):
        if not self.training or self.p == 0.0:
            return None
        pre = getattr(self, "_predrawn", None)
        if pre:                                  # drawn for the whole encoder in one go (draw_row_scales)
            return pre.pop()
        keep = 1.0 - self.p
        return torch.empty(n, dtype=torch.float32, device=device).bernoulli_(keep).div_(keep)


def draw_row_scales(model: nn.Module, n: int, device):
    """All StochasticDepth draws of one encoder forward in three launches instead of two tiny kernels per
    residual branch (46 per forward of Swin-T: each cost ~5 us on the critical path): one uniform draw
    [branches, n], one compare against each branch's keep probability, one scale.  Same distribution as the
    per-branch ``bernoulli_(keep).div_(keep)`` (1/keep with probability keep, else 0)."""
    mods = [m for m in model.modules() if isinstance(m, StochasticDepth) and m.training and m.p > 0.0]
    if not mods:
        return
    key = (str(device), tuple(m.p for m in mods))
    cached = getattr(model, "_sd_keep", None)
    if cached is None or cached[0] != key:       # built once: no host-to-device copy inside a captured step
        keep = torch.tensor([1.0 - m.p for m in mods for _ in range(2)], dtype=torch.float32).unsqueeze(1).to(device)
        model._sd_keep = cached = (key, keep)
    keep = cached[1]
    scales = (torch.rand(keep.shape[0], n, device=device) < keep).float() / keep
    for i, m in enumerate(mods):
        m._predrawn = [scales[2 * i + 1], scales[2 * i]]         # popped in call order: attention branch, then MLP


class ShiftedWindowAttention(nn.Module):
    def __init__(self, dim: int, window_size: List[int], shift_size: List[int], num_heads: int):
        super().__init__()
        self.window_size, self.shift_size, self.num_heads = window_size, shift_size, num_heads
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)
        n = (2 * window_size[0] - 1) * (2 * window_size[1] - 1)
        self.relative_position_bias_table = nn.Parameter(torch.zeros(n, num_heads))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        self.register_buffer("relative_position_index", _relative_position_index(window_size[0]))
        self._tab_key, self._tab = None, None

    def additive_table(self, shift: int, dtype, device, acc_order: bool = False) -> torch.Tensor:
        """[4][heads][64][64]: relative-position bias (swin_transformer.py:47-55) + the shift mask of the four
        window types (interior, last column, last row, corner; :190-203), PAD_LOGIT on the 15 pad keys.
        acc_order: the key columns of every 16-key group in MFMA accumulator-register order (mtmp_swin_attn_block)."""
        t = self.relative_position_bias_table
        key = (shift, dtype, t._version, t.data_ptr(), acc_order)
        if key != self._tab_key:
            L, h = WS * WS, self.num_heads
            bias = t.detach()[self.relative_position_index.long()].view(L, L, h).permute(2, 0, 1).float()
            tab = torch.zeros(4, h, 64, 64, dtype=torch.float32, device=device)
            tab[:, :, :, L:] = PAD_LOGIT
            tab[:, :, :L, :L] = bias
            if shift > 0:
                tab[:, :, :L, :L] += _shift_mask(2 * WS, 2 * WS, WS, shift, shift).to(device).unsqueeze(1)
            if acc_order:          # position 8 h + j of group g <- key 16 g + (j & 3) + 8 (j >> 2) + 4 h
                src = [16 * g + (j & 3) + 8 * (j >> 2) + 4 * h for g in range(4) for h in range(2) for j in range(8)]
                tab = tab[..., torch.tensor(src, device=device)]
            self._tab, self._tab_key = tab.to(dtype).contiguous(), key
        return self._tab

    def additive_table_train(self, shift: int, device) -> torch.Tensor:
        """The same [4][heads][64][64] table in fp32, built from relative_position_bias_table by differentiable torch ops: the
        gradient ops.WindowAttnFn returns for the table reaches the parameter through the index gather (swin_transformer.py:47-55)."""
        L, h = WS * WS, self.num_heads
        bias = self.relative_position_bias_table[self.relative_position_index.long()].view(L, L, h).permute(2, 0, 1).float()
        key = (shift, str(device))
        if getattr(self, "_base_key", None) != key:
            base = torch.zeros(4, h, 64, 64, dtype=torch.float32, device=device)
            base[:, :, :, L:] = PAD_LOGIT
            if shift > 0:
                base[:, :, :L, :L] += _shift_mask(2 * WS, 2 * WS, WS, shift, shift).to(device).unsqueeze(1)
            self._base, self._base_key = base, key
        return self._base + torch.nn.functional.pad(bias, (0, 64 - L, 0, 64 - L)).unsqueeze(0)

    def forward_train(self, xn: torch.Tensor) -> torch.Tensor:
        """xn [n,H,W,C] (normalised) -> attention output BEFORE the output projection, with gradients."""
        n, H, W, C = xn.shape
        if H % WS or W % WS:
            raise NotImplementedError("training the image encoder needs feature maps that are multiples of the 7x7 window "
                                      "(--image-size 224 / 448)")
        shift = 0 if WS >= H else self.shift_size[0]
        qkv = ops.LinearFn.apply(xn, self.qkv.weight, self.qkv.bias, xn.dtype)
        return ops.WindowAttnFn.apply(qkv, self.additive_table_train(shift, xn.device), self.num_heads, shift)

    def forward(self, xn: torch.Tensor, norm: nn.LayerNorm = None) -> torch.Tensor:
        """xn [n,H,W,C] (already normalised, or raw with `norm` = the block's norm1 to be fused into the qkv
        projection) -> attention output [n,H,W,C] BEFORE the output projection."""
        n, H, W, C = xn.shape
        Hp, Wp = -(-H // WS) * WS, -(-W // WS) * WS           # zero-padded to whole windows (:150-152)
        if (WS >= Hp) != (WS >= Wp):
            raise NotImplementedError("maps with one side of a single window and the other of several (per-axis shift)")
        shift = 0 if WS >= Hp else self.shift_size[0]
        if norm is not None:          # xn is the un-normalised map: norm1 + qkv in one launch (mtmp_swin_ln_linear)
            qkv = ops.swin_ln_linear(xn.view(-1, C), norm.weight, norm.bias, norm.eps, _w(self.qkv.weight, xn.dtype),
                                     self.qkv.bias).view(n, H, W, 3 * C)
        else:
            qkv = ops.gemm_nt(xn.view(-1, C), _w(self.qkv.weight, xn.dtype), self.qkv.bias).view(n, H, W, 3 * C)
        tab = self.additive_table(shift, xn.dtype, xn.device)
        if (Hp, Wp) == (H, W):
            return ops.swin_window_attn(qkv, tab, self.num_heads, shift)
        if ops.live_rows_active():
            raise NotImplementedError("present-images-only encoding needs maps that are multiples of the 7x7 window")
        # the pad tokens are zero vectors BEHIND norm1 (the reference pads the normalised map): q / k / v = the bias
        padded = self.qkv.bias.detach().to(qkv.dtype).expand(n, Hp, Wp, 3 * C).contiguous()
        padded[:, :H, :W] = qkv
        return ops.swin_window_attn(padded, tab, self.num_heads, shift)[:, :H, :W].contiguous()


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, window_size, shift_size, sd_prob):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn = ShiftedWindowAttention(dim, window_size, shift_size, num_heads)
        self.stochastic_depth = StochasticDepth(sd_prob, "row")
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        # torchvision MLP: Linear, GELU, Dropout, Linear, Dropout -> keys mlp.0 / mlp.3
        self.mlp = nn.Sequential(nn.Linear(dim, 4 * dim), nn.GELU(), nn.Dropout(0.0), nn.Linear(4 * dim, dim),
                                 nn.Dropout(0.0))
        for m in self.mlp.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.normal_(m.bias, std=1e-6)

    def draw_scales(self, n: int, device):
        """(attention-branch, MLP-branch) StochasticDepth factors of one forward, float32[n] each or None."""
        return self.stochastic_depth.row_scale(n, device), self.stochastic_depth.row_scale(n, device)

    def forward_train(self, x):
        """The block as autograd nodes (swin_transformer.py:428-449): x + sd(proj(attn(norm1 x))), then x + sd(mlp(norm2 x))."""
        n, H, W, C = x.shape
        dt = x.dtype
        s_attn, s_mlp = self.draw_scales(n, x.device)
        a = self.attn.forward_train(ops.LayerNormRowsFn.apply(x, self.norm1.weight, self.norm1.bias, self.norm1.eps))
        a = ops.LinearFn.apply(a, self.attn.proj.weight, self.attn.proj.bias, dt)
        if s_attn is not None:
            a = a * s_attn.view(n, 1, 1, 1).to(dt)
        x = x + a
        h = ops.LayerNormRowsFn.apply(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        h = ops.GeluFn.apply(ops.LinearFn.apply(h, self.mlp[0].weight, self.mlp[0].bias, dt))
        h = ops.LinearFn.apply(h, self.mlp[3].weight, self.mlp[3].bias, dt)
        if s_mlp is not None:
            h = h * s_mlp.view(n, 1, 1, 1).to(dt)
        return x + h

    def forward(self, x, scales=None):
        """x [n,H,W,C]; scales: this call's slice of draw_scales() when the batch is processed in parts (None = draw)."""
        n, H, W, C = x.shape
        dt, hw = x.dtype, H * W
        x2 = x.view(-1, C)
        s_attn, s_mlp = self.draw_scales(n, x.device) if scales is None else scales
        at = self.attn
        if dt == torch.bfloat16 and C in ops.SWIN_ATTN_BLOCK_WIDTHS and _FUSED_ATTN and H % WS == 0 and W % WS == 0:
            # stages 1-2: the whole attention half in one launch; the 3C-wide qkv map never exists
            shift = 0 if WS >= H else at.shift_size[0]
            x2 = ops.swin_attn_block(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, _w(at.qkv.weight, dt), at.qkv.bias,
                                     at.additive_table(shift, dt, x.device, acc_order=True), at.num_heads, shift,
                                     _w(at.proj.weight, dt), at.proj.bias, s_attn).view(-1, C)
        else:
            if dt == torch.bfloat16 and C in ops.SWIN_LN_LINEAR_WIDTHS and _FUSED_MLP:
                a = at(x, norm=self.norm1)
            else:
                a = at(ops.layernorm_rows(x, self.norm1.weight, self.norm1.bias, self.norm1.eps))
            x2 = ops.gemm_nt(a.view(-1, C), _w(at.proj.weight, dt), at.proj.bias, res2d=x2, row_scale=s_attn, rows_per_scale=hw)
        if dt == torch.bfloat16 and C in ops.SWIN_MLP_WIDTHS and _FUSED_MLP:
            # stages 1-2: norm2 -> fc1 -> GELU -> fc2 -> row scale -> residual in one launch; the 4C-wide hidden
            # activation (154 MB per stage-1 block of 64 images) stays in registers
            x2 = ops.swin_mlp(x2, self.norm2.weight, self.norm2.bias, self.norm2.eps, _w(self.mlp[0].weight, dt),
                              self.mlp[0].bias, _w(self.mlp[3].weight, dt), self.mlp[3].bias, s_mlp, hw)
            return x2.view(n, H, W, C)
        h = ops.layernorm_rows(x2, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        h = ops.gemm_nt(h, _w(self.mlp[0].weight, dt), self.mlp[0].bias, act="gelu")
        x2 = ops.gemm_nt(h, _w(self.mlp[3].weight, dt), self.mlp[3].bias, res2d=x2, row_scale=s_mlp, rows_per_scale=hw)
        return x2.view(n, H, W, C)


class PatchMerging(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim, eps=1e-5)

    def forward_train(self, x):
        n, H, W, C = x.shape
        if H % 2 or W % 2:
            x = torch.nn.functional.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        y = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)      # (:40-44)
        y = ops.LayerNormRowsFn.apply(y.contiguous(), self.norm.weight, self.norm.bias, self.norm.eps)
        return ops.LinearFn.apply(y, self.reduction.weight, None, x.dtype)

    def forward(self, x):
        n, H, W, C = x.shape
        if H % 2 or W % 2:            # _patch_merging_pad (:34-44): a zero row / column behind the map
            if ops.live_rows_active():
                raise NotImplementedError("present-images-only encoding needs even-sized maps in front of every patch merging")
            x = torch.nn.functional.pad(x, (0, 0, 0, W % 2, 0, H % 2))
            H, W = H + H % 2, W + W % 2
        y = ops.layernorm_rows(x, self.norm.weight, self.norm.bias, self.norm.eps, merge_hw=(H, W))
        return ops.gemm_nt(y.view(-1, 4 * C), _w(self.reduction.weight, x.dtype)).view(n, H // 2, W // 2, 2 * C)


class SwinTransformer(nn.Module):
    def __init__(self, patch_size=(4, 4), embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24),
                 window_size=(7, 7), stochastic_depth_prob=0.2, num_classes=1000,
                 compute_dtype: torch.dtype = torch.bfloat16):
        super().__init__()
        self.compute_dtype = compute_dtype
        layers: List[nn.Module] = [nn.Sequential(
            nn.Conv2d(1, embed_dim, kernel_size=tuple(patch_size), stride=tuple(patch_size)),
            Permute([0, 2, 3, 1]), nn.LayerNorm(embed_dim, eps=1e-5))]
        total, bid = sum(depths), 0
        for si, depth in enumerate(depths):
            dim = embed_dim * 2 ** si
            blocks = []
            for bi in range(depth):
                sd = stochastic_depth_prob * float(bid) / (total - 1)
                blocks.append(SwinTransformerBlock(dim, num_heads[si], list(window_size),
                                                   [0 if bi % 2 == 0 else w // 2 for w in window_size], sd))
                bid += 1
            layers.append(nn.Sequential(*blocks))
            if si < len(depths) - 1:
                layers.append(PatchMerging(dim))
        self.features = nn.Sequential(*layers)
        nf = embed_dim * 2 ** (len(depths) - 1)
        self.norm = nn.LayerNorm(nf, eps=1e-5)
        self.permute = Permute([0, 3, 1, 2])
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.flatten = nn.Flatten(1)
        self.head = nn.Linear(nf, num_classes)          # never evaluated (reference :611-618), kept for the state_dict
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def predraw(self, n, device):
        """The StochasticDepth draws of the next forward(x) with x.shape[0] == n, issued on the CURRENT stream; forward() waits
        for them behind its first kernel (a caller that runs the encoder on a stream of its own gets the draws off that stream)."""
        draw_row_scales(self, n, device)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._predrawn_ev = (ev, n)

    def trains(self) -> bool:
        return torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

    def forward_train(self, x):
        """x [B,1,H,W] fp32 -> [B,H/32,W/32,768] with gradients for every encoder parameter (module docstring): the 4x4/4 patch
        embedding as a projection of the 16-pixel patches (:559-567), the blocks, the patch mergings, the final norm."""
        stem = self.features[0]
        n, _, H, W = x.shape
        dt = self.compute_dtype
        draw_row_scales(self, n, x.device)
        ph, pw = stem[0].kernel_size
        patches = x.reshape(n, H // ph, ph, W // pw, pw).permute(0, 1, 3, 2, 4).reshape(n, H // ph, W // pw, ph * pw)
        y = ops.LinearFn.apply(patches.to(dt), stem[0].weight.view(stem[0].weight.shape[0], -1), stem[0].bias, dt)
        y = ops.LayerNormRowsFn.apply(y, stem[2].weight, stem[2].bias, stem[2].eps)
        for layer in list(self.features)[1:]:
            if isinstance(layer, nn.Sequential):
                for blk in layer:
                    y = blk.forward_train(y)
            else:
                y = layer.forward_train(y)
        return ops.LayerNormRowsFn.apply(y, self.norm.weight, self.norm.bias, self.norm.eps)

    def forward(self, x, tail_streams=None, slots=None):
        """x [B,1,H,W] fp32 -> [B,H/32,W/32,768] in the compute dtype.

        slots = ops.image_slots(present, (H/4)(W/4)): only the PRESENT images are encoded (the reference pushes a zero image
        through the whole encoder for a sample without one, and nothing reads the result: the bottleneck exchange gives the
        image stream weight 0 for it).  The present images are processed in the first slots of every buffer, every launch
        reads its live row count from the slot table (ops.rows_live), and the result comes back in batch order with ZEROS
        for the samples without an image.  Shapes and grids stay those of the whole batch (hipGraph replay).  A sample's
        StochasticDepth draws are those of its slot.

        tail_streams=(s0, s1): two HIP streams OTHER than the caller's.  Stages 3-4 (12.5 k / 3.1 k tokens for 64 images:
        launches of 15-35 us that cannot fill 256 CUs and are bound by one workgroup's latency chain) then run as two half
        batches side by side on s0 and s1, s0 joins s1, and the result is valid ON s0: the caller goes on there.  (The
        halves cannot come back to the caller's stream: under hipGraph capture this ROCm crashes in hipStreamEndCapture
        when a forked non-origin stream is joined by the stream that forked it -- tools/dbg/capture_topology.py.)"""
        if slots is None and self.trains():
            return self.forward_train(x)
        stem = self.features[0]
        n = x.shape[0]
        drawn = getattr(self, "_predrawn_ev", None)
        if drawn is None:
            draw_row_scales(self, n, x.device)
        # the slot table is read through raw pointers by launches on this stream AND on the tail streams: it must outlive all of
        # them (the caching allocator only knows the stream it was made on), so this module keeps it until the next forward
        self._slots_keepalive = slots
        live = (lambda part, s: ops.rows_live(slots, 2 * n + 1 + 5 * part + s)) if slots is not None else \
               (lambda part, s: contextlib.nullcontext())
        res_of = [0, 1, 1, 2, 2, 3, 3]        # resolution index (rows per image = (H/4)(W/4) >> 2 r) of each feature layer's OUTPUT
        with live(0, 0):
            x = ops.swin_stem(x, stem[0].weight, stem[0].bias, stem[2].weight, stem[2].bias, self.compute_dtype, order=slots)
        if drawn is not None:                 # predraw(): the draws were issued on another stream, beside the patch embedding
            if drawn[1] != n:
                raise RuntimeError("SwinTransformer.predraw() was made for %d images, forward() got %d" % (drawn[1], n))
            torch.cuda.current_stream().wait_event(drawn[0])
            self._predrawn_ev = None
        layers = list(self.features)[1:]
        split = tail_streams is not None and _SPLIT_TAIL and n >= 16 and n % 2 == 0 and len(layers) == 7
        if slots is not None and len(layers) != 7:
            raise NotImplementedError("slots: Swin-T layout (four stages) only")

        def finish(out_ext):
            """[n + 1, ...] with the encoder's result in slots 0..n-1 -> batch order, zeros for samples without an image"""
            out_ext[n].zero_()
            return out_ext.index_select(0, slots[n:2 * n])
        for k, layer in enumerate(layers[:4] if split else layers):
            with live(0, res_of[k] if slots is not None else 0):
                x = layer(x)
        if split:
            tail = []                 # (module, per-block scale pairs) in execution order; all draws happen once, here
            for layer in layers[4:]:
                if isinstance(layer, nn.Sequential):
                    tail += [(blk, blk.draw_scales(n, x.device)) for blk in layer]
                else:
                    tail.append((layer, None))
            head = torch.cuda.current_stream()
            s0, s1 = tail_streams
            s0.wait_stream(head)
            s1.wait_stream(head)
            if slots is not None and not torch.cuda.is_current_stream_capturing():
                slots.record_stream(s0)
                slots.record_stream(s1)
            with torch.cuda.stream(s0):       # s0 owns the result: nothing allocated on s1 is read by another stream
                out = torch.empty(n + (slots is not None), x.shape[1] // 2, x.shape[2] // 2, 2 * x.shape[3], dtype=x.dtype,
                                  device=x.device)   # one merge left
            # (Under rocprofv3 --kernel-trace the second half starts ~600 us after the first, as if serialised; the step's own
            #  one-lane mark kernels -- tools/dbg/timeline.py, no profiler attached -- show both halves starting within 0.1 us
            #  of each other and ending 12 us apart.  Issuing the halves module by module, alternating streams, changes nothing.)
            for k, st in enumerate((s0, s1)):
                lo, hi = k * n // 2, (k + 1) * n // 2
                with torch.cuda.stream(st):
                    ops.mark("swin.tail%d.s" % k)          # (nothing is launched unless marks are enabled)
                    y = x[lo:hi]
                    for j, (mod, sc) in enumerate(tail):
                        with live(1 + k, 3 if j >= 6 else 2):      # (six stage-3 blocks; the merge and the two stage-4 blocks write the coarser map)
                            y = mod(y) if sc is None else mod(y, tuple(None if v is None else v[lo:hi] for v in sc))
                    with live(1 + k, 3):
                        ops.layernorm_rows(y, self.norm.weight, self.norm.bias, self.norm.eps, out=out[lo:hi])
                    ops.mark("swin.tail%d.e" % k)
            s0.wait_stream(s1)
            if slots is not None:
                with torch.cuda.stream(s0):
                    return finish(out)
            return out
        if slots is None:
            return ops.layernorm_rows(x, self.norm.weight, self.norm.bias, self.norm.eps)
        out = torch.empty(n + 1, *x.shape[1:], dtype=x.dtype, device=x.device)
        with live(0, 3):
            ops.layernorm_rows(x, self.norm.weight, self.norm.bias, self.norm.eps, out=out[:n])
        return finish(out)


def swin_t_m(*, weights=None, progress: bool = True, compute_dtype=torch.bfloat16, **kwargs) -> SwinTransformer:
    """Swin-T with the reference's hyper-parameters (swin_transformer.py:835-842).  `weights` is accepted
    for signature parity; pretrained tensors are loaded through load_state_dict (no network here)."""
    return SwinTransformer(compute_dtype=compute_dtype, **kwargs)
