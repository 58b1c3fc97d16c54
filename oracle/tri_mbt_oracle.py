"""CPU ORACLE (test infrastructure -- NOT product code).

Plain PyTorch fp32 restatement of the reference's tri-modal training hot path,
written as pure functions over a ``state_dict`` (keys identical to the
reference module's) so the same weights can be pushed through the real
reference (in the build container), through this oracle, and through the HIP
product path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this file.  The product package never does.

Parity pin: the reference ships no tests and no golden vectors,
so this oracle is pinned against outputs of the reference itself, generated in
the build container by ``tests/golden/gen/make_golden.py`` (which imports
``/root/reference`` directly) and committed as ``tests/golden/*.npz``.
``tests/test_oracle_golden.py`` checks this file against those vectors.

Every function cites the reference lines it restates (paths relative to the
reference root).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

MASK_FILL = -65504.0  # builder/models/src/transformer/attention.py:38
N_BOTTLENECK = 4      # builder/models/8_missing_models/tri_mbt_vsltcls.py:37


# --------------------------------------------------------------------------
# small building blocks
# --------------------------------------------------------------------------
def custom_layernorm(z: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                     eps: float = 1e-6) -> torch.Tensor:
    """module.py:138-144 -- unbiased std, eps added to the std (not the var)."""
    mu = z.mean(dim=-1, keepdim=True)
    sd = z.std(dim=-1, keepdim=True)          # Bessel-corrected, like torch.std
    return gamma * ((z - mu) / (sd + eps)) + beta


def lin_ln_relu(sd: SD, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """nn.Sequential(Linear, nn.LayerNorm(eps 1e-5), ReLU) -- tri_mbt_vsltcls.py:59-76."""
    y = F.linear(x, sd[prefix + ".0.weight"], sd[prefix + ".0.bias"])
    y = F.layer_norm(y, (y.shape[-1],), sd[prefix + ".1.weight"], sd[prefix + ".1.bias"], 1e-5)
    return torch.relu(y)


def sinusoid_table(length: int, d_model: int) -> torch.Tensor:
    """module.py:21-32 (PositionalEncoding), rows [0, length)."""
    pos = torch.arange(0, length, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
    pe = torch.zeros(length, d_model)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def key_pad_mask(n_tokens: int, kv_len: torch.Tensor) -> torch.Tensor:
    """utils.py:79-125 -- bool [B, N, N]; True where key j >= kv_len[b]."""
    j = torch.arange(n_tokens).view(1, 1, n_tokens)
    m = j >= kv_len.view(-1, 1, 1).to(torch.long)
    return m.expand(-1, n_tokens, -1)


def multi_head_attention(sd: SD, prefix: str, x: torch.Tensor, mask: Optional[torch.Tensor],
                         n_head: int) -> torch.Tensor:
    """attention.py:24-84 -- head-major [H*B, N, dh] batches, score/sqrt(dh),
    masked_fill(-65504), softmax, no output projection, no attention dropout."""
    B, N, D = x.shape
    dh = D // n_head

    def proj(name):
        y = F.linear(x, sd[f"{prefix}.{name}_proj.linear.weight"], sd[f"{prefix}.{name}_proj.linear.bias"])
        y = y.view(B, N, n_head, dh).permute(2, 0, 1, 3).contiguous().view(n_head * B, N, dh)
        return y

    q, k, v = proj("query"), proj("key"), proj("value")
    ctx = scaled_dot_attention_headmajor(q, k, v, mask, n_head)
    ctx = ctx.view(n_head, B, N, dh).permute(1, 2, 0, 3).contiguous().view(B, N, D)
    return ctx


def scaled_dot_attention_headmajor(q, k, v, mask, n_head):
    """attention.py:24-49 on head-major [H*B, N, dh] batches."""
    dh = q.shape[-1]
    score = torch.bmm(q, k.transpose(1, 2)) / float(math.sqrt(dh))
    if mask is not None:
        score = score.masked_fill(mask.repeat(n_head, 1, 1), MASK_FILL)
    attn = torch.softmax(score, dim=-1)
    return torch.bmm(attn, v)


def attention_core(qkv: torch.Tensor, kv_len: Optional[torch.Tensor], n_head: int = 4) -> torch.Tensor:
    """The attention of attention.py:65-84 given already-projected qkv [B,N,3*D] (q|k|v): head split to
    [H*B,N,dh], masked softmax attention, head merge -> [B,N,D].  kv_len None = no mask."""
    B, N, D3 = qkv.shape
    D = D3 // 3
    dh = D // n_head

    def heads(t):
        return t.reshape(B, N, n_head, dh).permute(2, 0, 1, 3).contiguous().view(n_head * B, N, dh)

    mask = None if kv_len is None else key_pad_mask(N, kv_len)
    ctx = scaled_dot_attention_headmajor(heads(qkv[..., :D]), heads(qkv[..., D:2 * D]), heads(qkv[..., 2 * D:]),
                                         mask, n_head)
    return ctx.view(n_head, B, N, dh).permute(1, 2, 0, 3).contiguous().view(B, N, D)


def ffn_conv1x1(sd: SD, prefix: str, x: torch.Tensor, dropout_p: float = 0.0,
                training: bool = False) -> torch.Tensor:
    """module.py:74-80 -- Conv1d(k=1) -> ReLU -> drop -> Conv1d(k=1) -> drop."""
    w1, b1 = sd[prefix + ".w_1.weight"], sd[prefix + ".w_1.bias"]   # [d_ff, d, 1]
    w2, b2 = sd[prefix + ".w_2.weight"], sd[prefix + ".w_2.bias"]   # [d, d_ff, 1]
    h = torch.relu(F.conv1d(x.transpose(1, 2), w1, b1))
    h = F.dropout(h, dropout_p, training)
    y = F.conv1d(h, w2, b2).transpose(1, 2)
    return F.dropout(y, dropout_p, training)


def encoder_layer(sd: SD, prefix: str, x: torch.Tensor, mask: Optional[torch.Tensor], n_head: int,
                  dropout_p: float = 0.0, training: bool = False) -> torch.Tensor:
    """encoder.py:23-34 -- pre-LN residual block."""
    h = custom_layernorm(x, sd[prefix + ".attention_prenorm.gamma"], sd[prefix + ".attention_prenorm.beta"])
    x = multi_head_attention(sd, prefix + ".self_attention", h, mask, n_head) + x
    h = custom_layernorm(x, sd[prefix + ".feed_forward_prenorm.gamma"], sd[prefix + ".feed_forward_prenorm.beta"])
    return ffn_conv1x1(sd, prefix + ".feed_forward", h, dropout_p, training) + x


# --------------------------------------------------------------------------
# key-valid lengths (bit-exact integer artefacts)
# --------------------------------------------------------------------------
def fusion_kv_lengths(input_lengths: torch.Tensor, txt_lengths: torch.Tensor,
                      img_time: Optional[torch.Tensor], multiimages: int,
                      n_img_tokens: int) -> List[Optional[torch.Tensor]]:
    """Valid-key counts inside a fusion layer, per stream, sequence order
    [4 bottleneck | CLS | tokens] (mbt_encoder.py:745).

    vslt : 4 + 1 + input_lengths                      (tri_mbt_vsltcls.py:237, mbt_encoder.py:704,748)
    img  : None (unmasked) when multiimages == 0      (tri_mbt_vsltcls.py:124-127,144)
           4 + 1 + 49*#{k: img_time[b,k] != 10} else  (tri_mbt_vsltcls.py:226-232)
    txt  : L = txt_lengths + 2 + 1; L == 3 -> 0; then 4 + L  (mbt_encoder.py:704-707,748)
    """
    v = input_lengths.to(torch.long) + 1 + N_BOTTLENECK
    t = txt_lengths.to(torch.long) + 2 + 1
    t = torch.where(t == 3, torch.zeros_like(t), t) + N_BOTTLENECK
    if multiimages == 1:
        cnt = torch.count_nonzero(img_time.reshape(input_lengths.shape[0], -1) - 10, dim=1).to(torch.long)
        i = cnt * n_img_tokens + 1 + N_BOTTLENECK
    else:
        i = None
    return [v, i, t]


def prefusion_kv_lengths(input_lengths, txt_lengths, img_time, multiimages, n_img_tokens):
    """Same, for layers below fusion_startidx (no bottleneck prefix) -- mbt_encoder.py:703-714."""
    out = fusion_kv_lengths(input_lengths, txt_lengths, img_time, multiimages, n_img_tokens)
    return [None if o is None else o - N_BOTTLENECK for o in out]


def missing_to_num(missing: torch.Tensor) -> torch.Tensor:
    """trainer.py:67-77 -- rows (vslt,img,txt) 0/1 -> {0 full, 1 txt missing, 2 img missing, 3 both}
    through torch.unique(dim=0, return_inverse) over [4 template rows; batch rows]."""
    tmpl = torch.tensor([[0., 0., 0.], [0., 0., 1.], [0., 1., 0.], [0., 1., 1.]])
    _, inv = torch.unique(torch.cat([tmpl, missing.float()], 0), dim=0, sorted=True, return_inverse=True)
    return inv[4:].to(torch.long)


# --------------------------------------------------------------------------
# MBT fusion encoder -- mbt_encoder.py:696-784
# --------------------------------------------------------------------------
def mbt_encoder(sd: SD, prefix: str, streams: Sequence[torch.Tensor],
                input_lengths: torch.Tensor, txt_lengths: torch.Tensor, img_time,
                missing_num: torch.Tensor, *, n_layers: int, n_head: int,
                fusion_startidx: int = 0, vsltonly: int = 1, resbottle: bool = False,
                multiimages: int = 0, use_pe=(False, False, True),
                dropout_p: float = 0.0, training: bool = False):
    """Returns the list of per-stream outputs (length 1 when the vslt-only last
    layer short-circuits, mbt_encoder.py:757-763) *including* the CLS row and
    excluding the bottleneck rows, plus the final bottlenecks."""
    B = streams[0].shape[0]
    d = streams[0].shape[-1]
    # tokens per image: the reference hard-codes 3 images x 49 tokens (tri_mbt_vsltcls.py:226-231)
    n_img_tok = streams[1].shape[1] // (img_time.numel() // B) if multiimages == 1 else streams[1].shape[1]
    p = prefix + "." if prefix else ""
    # CLS prepend (:697-699)
    xs = [torch.cat([sd[f"{p}cls_token_per_modality.{m}"].expand(B, 1, d), s], 1) for m, s in enumerate(streams)]
    kv_f = fusion_kv_lengths(input_lengths, txt_lengths, img_time, multiimages, n_img_tok)
    kv_p = prefusion_kv_lengths(input_lengths, txt_lengths, img_time, multiimages, n_img_tok)
    # stream input norm (+PE) + dropout (:719-729)
    ys = []
    for m, x in enumerate(xs):
        y = F.layer_norm(x, (d,), sd[f"{p}layer_norms_in.{m}.weight"], sd[f"{p}layer_norms_in.{m}.bias"], 1e-5)
        if use_pe[m]:
            y = y + sinusoid_table(x.shape[1], d).unsqueeze(0)
        ys.append(F.dropout(y, dropout_p, training))
    bott = sd[f"{p}bottlenecks"].expand(B, N_BOTTLENECK, d)
    idx = torch.arange(B)
    for layer in range(n_layers):
        xs, ys = ys, []
        if layer < fusion_startidx:                                            # (:734-737)
            for m in range(3):
                mask = None if kv_p[m] is None else key_pad_mask(xs[m].shape[1], kv_p[m])
                ys.append(encoder_layer(sd, f"{p}layer_stacks.{layer}.{m}", xs[m], mask, n_head, dropout_p, training))
            continue
        b_out = []
        res = bott
        last = (layer + 1 == n_layers) and vsltonly == 1
        for m in range(3):
            z = torch.cat([bott, xs[m]], 1)                                    # (:745)
            mask = None if kv_f[m] is None else key_pad_mask(z.shape[1], kv_f[m])
            o = encoder_layer(sd, f"{p}layer_stacks.{layer}.{m}", z, mask, n_head, dropout_p, training)
            b_out.append(o[:, :N_BOTTLENECK])
            ys.append(o[:, N_BOTTLENECK:])
            if last:
                break
        if last:
            break
        st = torch.stack(b_out)                                                # [3,B,4,d]  (:764-768)
        cand = torch.stack([st.mean(0), st[:2].mean(0), torch.stack([st[0], st[2]]).mean(0), st[0]])
        bott = cand[missing_num, idx]                                          # (:776)
        if resbottle:
            bott = torch.stack([bott, res]).mean(0)                            # (:778-779)
    return ys, bott


def mbt_encoder_bimodal(sd: SD, prefix: str, streams: Sequence[torch.Tensor], varying_lengths: Sequence[torch.Tensor],
                        missing: torch.Tensor, *, n_layers: int, n_head: int, txt_idx: int = 2,
                        use_pe=(True, True), mask=(True, True), dropout_p: float = 0.0, training: bool = False):
    """BimodalTransformerEncoder_MBT.forward (mbt_encoder.py:575-634): two streams, EVERY layer is a fusion layer
    (the uni-modal branch is commented out, :610-615), bottleneck candidates = (mean of both, stream 0 alone) picked
    by ``missing`` in {0, 1} (:629-632).  ``varying_lengths`` as the caller passes them (before the +1 of :584)."""
    B, d = streams[0].shape[0], streams[0].shape[-1]
    p = prefix + "." if prefix else ""
    xs = [torch.cat([sd[f"{p}cls_token_per_modality.{m}"].expand(B, 1, d), s], 1) for m, s in enumerate(streams)]
    kv = []
    for m in range(2):
        v = varying_lengths[m].clone() + 1                                     # (:584)
        if m == txt_idx:
            v[v == 3] = 0                                                      # (:586-587)
        kv.append(v + N_BOTTLENECK if mask[m] else None)                       # (:621)
    ys = []
    for m, x in enumerate(xs):                                                 # (:596-607)
        y = F.layer_norm(x, (d,), sd[f"{p}layer_norms_in.{m}.weight"], sd[f"{p}layer_norms_in.{m}.bias"], 1e-5)
        if use_pe[m]:
            y = y + sinusoid_table(x.shape[1], d).unsqueeze(0)
        ys.append(F.dropout(y, dropout_p, training))
    bott = sd[f"{p}bottlenecks"].expand(B, N_BOTTLENECK, d)
    idx = torch.arange(B)
    for layer in range(n_layers):
        xs, ys, b_out = ys, [], []
        for m in range(2):
            z = torch.cat([bott, xs[m]], 1)                                    # (:618)
            km = None if kv[m] is None else key_pad_mask(z.shape[1], kv[m])
            o = encoder_layer(sd, f"{p}layer_stacks.{layer}.{m}", z, km, n_head, dropout_p, training)
            b_out.append(o[:, :N_BOTTLENECK])
            ys.append(o[:, N_BOTTLENECK:])
        st = torch.stack(b_out)
        bott = torch.stack([st.mean(0), st[0]])[missing, idx]                  # (:629-632)
    return ys, bott


# --------------------------------------------------------------------------
# Swin-T forward (eval; frozen, no_grad in the model) -- swin_transformer.py
# --------------------------------------------------------------------------
SWIN_DEPTHS = (2, 2, 6, 2)
SWIN_HEADS = (3, 6, 12, 24)
SWIN_WS = 7


def swin_relative_position_index(ws: int = SWIN_WS) -> torch.Tensor:
    """swin_transformer.py:263-275."""
    ch = torch.arange(ws)
    coords = torch.stack(torch.meshgrid(ch, ch, indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1).flatten()


def swin_shift_mask(Hp: int, Wp: int, ws: int, shift_h: int, shift_w: int) -> torch.Tensor:
    """swin_transformer.py:190-203 -- [nW, ws*ws, ws*ws] of 0 / -100."""
    img = torch.zeros(Hp, Wp)
    cuts_h = ((0, -ws), (-ws, -shift_h), (-shift_h, None))
    cuts_w = ((0, -ws), (-ws, -shift_w), (-shift_w, None))
    c = 0
    for h in cuts_h:
        for w in cuts_w:
            img[h[0]:h[1], w[0]:w[1]] = c
            c += 1
    img = img.view(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    diff = img.unsqueeze(1) - img.unsqueeze(2)
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


def swin_window_attention(sd: SD, prefix: str, x: torch.Tensor, heads: int, shift: int) -> torch.Tensor:
    """swin_transformer.py:115-225 (V1 branch: q scaled by dh**-0.5, +rel-pos bias, -100 shift mask)."""
    B, H, W, C = x.shape
    ws = SWIN_WS
    pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
    x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))
    Hp, Wp = x.shape[1], x.shape[2]
    sh = [shift, shift]
    if ws >= Hp:
        sh[0] = 0
    if ws >= Wp:
        sh[1] = 0
    if sum(sh) > 0:
        x = torch.roll(x, shifts=(-sh[0], -sh[1]), dims=(1, 2))
    nW = (Hp // ws) * (Wp // ws)
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B * nW, ws * ws, C)
    qkv = F.linear(x, sd[prefix + ".qkv.weight"], sd[prefix + ".qkv.bias"])
    qkv = qkv.reshape(B * nW, ws * ws, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (C // heads) ** -0.5, qkv[1], qkv[2]
    attn = q.matmul(k.transpose(-2, -1))
    idx = sd.get(prefix + ".relative_position_index", None)
    if idx is None:
        idx = swin_relative_position_index(ws)
    bias = sd[prefix + ".relative_position_bias_table"][idx.long()].view(ws * ws, ws * ws, -1).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if sum(sh) > 0:
        m = swin_shift_mask(Hp, Wp, ws, sh[0], sh[1])
        attn = attn.view(B, nW, heads, ws * ws, ws * ws) + m.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, heads, ws * ws, ws * ws)
    attn = torch.softmax(attn, dim=-1)
    x = attn.matmul(v).transpose(1, 2).reshape(B * nW, ws * ws, C)
    x = F.linear(x, sd[prefix + ".proj.weight"], sd[prefix + ".proj.bias"])
    x = x.view(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
    if sum(sh) > 0:
        x = torch.roll(x, shifts=(sh[0], sh[1]), dims=(1, 2))
    return x[:, :H, :W, :].contiguous()


def swin_forward(sd: SD, prefix: str, img: torch.Tensor, row_scales=None) -> torch.Tensor:
    """swin_transformer.py:559-618 + 621-654 (1-channel 4x4/4 stem): [B,1,224,224] -> [B,7,7,768] (final LayerNorm,
    no pooling/head).  row_scales = None: eval mode.  Train mode (2_train.py:128 puts the frozen encoder back into it):
    row_scales is a list of 12 (attention-branch, MLP-branch) pairs of float[B] factors -- the row-mode
    StochasticDepth draws of swin_transformer.py:437,448-449 (torchvision.ops.stochastic_depth: each residual branch
    of a sample is multiplied by bernoulli(1-p)/(1-p)), injected so that a test can give both sides the same draws."""
    p = prefix + "."
    scales = iter(row_scales) if row_scales is not None else None
    bc = lambda v: v.view(-1, 1, 1, 1)
    x = F.conv2d(img, sd[p + "features.0.0.weight"], sd[p + "features.0.0.bias"], stride=4)
    x = x.permute(0, 2, 3, 1)
    x = F.layer_norm(x, (x.shape[-1],), sd[p + "features.0.2.weight"], sd[p + "features.0.2.bias"], 1e-5)
    for stage, (depth, heads) in enumerate(zip(SWIN_DEPTHS, SWIN_HEADS)):
        fi = 1 + 2 * stage
        for blk in range(depth):
            bp = f"{p}features.{fi}.{blk}"
            C = x.shape[-1]
            s_attn, s_mlp = next(scales) if scales is not None else (None, None)
            h = F.layer_norm(x, (C,), sd[bp + ".norm1.weight"], sd[bp + ".norm1.bias"], 1e-5)
            a = swin_window_attention(sd, bp + ".attn", h, heads, 0 if blk % 2 == 0 else SWIN_WS // 2)
            x = x + (a if s_attn is None else bc(s_attn) * a)
            h = F.layer_norm(x, (C,), sd[bp + ".norm2.weight"], sd[bp + ".norm2.bias"], 1e-5)
            h = F.gelu(F.linear(h, sd[bp + ".mlp.0.weight"], sd[bp + ".mlp.0.bias"]))
            f = F.linear(h, sd[bp + ".mlp.3.weight"], sd[bp + ".mlp.3.bias"])
            x = x + (f if s_mlp is None else bc(s_mlp) * f)
        if stage < len(SWIN_DEPTHS) - 1:                                    # PatchMerging :60-85
            mp = f"{p}features.{fi + 1}"
            Hh, Ww = x.shape[1], x.shape[2]
            x = F.pad(x, (0, 0, 0, Ww % 2, 0, Hh % 2))
            x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
            x = F.layer_norm(x, (x.shape[-1],), sd[mp + ".norm.weight"], sd[mp + ".norm.bias"], 1e-5)
            x = F.linear(x, sd[mp + ".reduction.weight"])
    return F.layer_norm(x, (x.shape[-1],), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5)


# --------------------------------------------------------------------------
# full model forward -- tri_mbt_vsltcls.py:167-263 (TIE / biobert / swin path)
# --------------------------------------------------------------------------
def tie_embedding(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """tri_mbt_vsltcls.py:183-190 -- x[B,T,3] = (time, value, feature index)."""
    val = lin_ln_relu(sd, "ie_vslt", x[:, :, 1].unsqueeze(2))
    tim = lin_ln_relu(sd, "ie_time", x[:, :, 0].unsqueeze(2))
    feat = F.embedding(x[:, :, 2].to(torch.int32).long(), sd["ie_feat.weight"])
    return val + tim + feat


class Cfg:
    """The subset of control/config.py flags that the hot path reads."""

    def __init__(self, n_layers=6, n_head=4, d_model=256, fusion_startidx=0, vsltonly=1,
                 resbottle=0, multiimages=0, imgtxt_time=1, dropout=0.0):
        self.n_layers, self.n_head, self.d_model = n_layers, n_head, d_model
        self.fusion_startidx, self.vsltonly, self.resbottle = fusion_startidx, vsltonly, resbottle
        self.multiimages, self.imgtxt_time, self.dropout = multiimages, imgtxt_time, dropout


def model_forward(sd: SD, cfg: Cfg, x, age, gen, input_lengths, txts, txt_lengths, img, missing_num,
                  img_time, txt_time, training: bool = False, swin_features: Optional[torch.Tensor] = None):
    """Returns logits [B,1].  ``training`` selects BatchNorm batch statistics
    (fc_list.1) and dropout (with cfg.dropout); Swin always runs in eval, under
    no_grad, as tri_mbt_vsltcls.py:104,208-209 intends (see DESIGN.md on
    StochasticDepth)."""
    B = x.shape[0]
    demo = lin_ln_relu(sd, "ie_demo", torch.stack([age, gen], 1))                          # :176-177
    v = tie_embedding(sd, x)                                                               # :183-190
    t = F.linear(txts, sd["txt_embedding.weight"], sd["txt_embedding.bias"])               # :200
    if swin_features is None:
        with torch.no_grad():                                                              # :205-209
            imgs = img.reshape(-1, 1, img.shape[-2], img.shape[-1])
            swin_features = swin_forward(sd, "img_encoder", imgs)
    i = F.linear(swin_features.flatten(1, 2), sd["linear.weight"], sd["linear.bias"])      # :210-211
    img_time_flat = img_time.reshape(-1)
    if cfg.imgtxt_time == 1:                                                               # :216-224
        i = i + lin_ln_relu(sd, "ie_time", img_time_flat.unsqueeze(1)).unsqueeze(1) + sd["ie_feat.weight"][18]
        t = t + lin_ln_relu(sd, "ie_time", txt_time.unsqueeze(1)).unsqueeze(1) + sd["ie_feat.weight"][19]
    if cfg.multiimages == 1:                                                               # :226-232
        i = i.reshape(B, -1, cfg.d_model)
    outs, _ = mbt_encoder(sd, "fusion_transformer", [v, i, t], input_lengths, txt_lengths, img_time_flat,
                          missing_num, n_layers=cfg.n_layers, n_head=cfg.n_head,
                          fusion_startidx=cfg.fusion_startidx, vsltonly=cfg.vsltonly,
                          resbottle=bool(cfg.resbottle), multiimages=cfg.multiimages,
                          dropout_p=cfg.dropout, training=training)
    cls = F.layer_norm(outs[0][:, 0, :], (cfg.d_model,), sd["layer_norms_after_concat.weight"],
                       sd["layer_norms_after_concat.bias"], 1e-5)                          # :248
    h = torch.cat([cls, demo], 1)
    h = F.linear(h, sd["fc_list.0.weight"], sd["fc_list.0.bias"])
    if training:
        # nn.BatchNorm1d in train mode: batch statistics, running buffers updated in place
        h = F.batch_norm(h, sd["fc_list.1.running_mean"], sd["fc_list.1.running_var"],
                         sd["fc_list.1.weight"], sd["fc_list.1.bias"], True, 0.1, 1e-5)
    else:
        h = F.batch_norm(h, sd["fc_list.1.running_mean"], sd["fc_list.1.running_var"],
                         sd["fc_list.1.weight"], sd["fc_list.1.bias"], False, 0.1, 1e-5)
    h = torch.relu(h)
    return F.linear(h, sd["fc_list.3.weight"], sd["fc_list.3.bias"])                       # :253-255


def bi_vslttxt_forward(sd: SD, x, age, gen, input_lengths, txts, txt_lengths, missing, txt_time, *, n_layers: int,
                       n_head: int = 4, imgtxt_time: int = 1, dropout: float = 0.0, training: bool = False):
    """BI_VSLTTXT_MBT_V1.forward (8_missing_models/bi_vslttxt_mbt_v1.py:121-183, TIE / biobert path): logits [B].
    ``missing`` in {0, 1} as the trainer hands it over for --input-types vslt_txt (trainer.py:99-101)."""
    d = sd["ie_feat.weight"].shape[1]
    demo = lin_ln_relu(sd, "ie_demo", torch.stack([age, gen], 1))                          # :129-130,138-146
    v = tie_embedding(sd, x)                                                               # :139-145
    t = F.linear(txts, sd["txt_embedding.weight"], sd["txt_embedding.bias"])               # :157
    if imgtxt_time == 1:                                                                   # :159-165
        t = t + lin_ln_relu(sd, "ie_time", txt_time.unsqueeze(1)).unsqueeze(1) + sd["ie_feat.weight"][19]
    outs, _ = mbt_encoder_bimodal(sd, "fusion_transformer", [v, t], [input_lengths, txt_lengths + 2], missing,
                                  n_layers=n_layers, n_head=n_head, txt_idx=1, use_pe=(False, True), mask=(True, True),
                                  dropout_p=dropout, training=training)
    c = torch.stack([outs[0][:, 0, :], outs[1][:, 0, :]])                                  # :171-174
    cls = torch.stack([c.mean(0), outs[0][:, 0, :]])[missing, torch.arange(x.shape[0])]
    cls = F.layer_norm(cls, (d,), sd["layer_norms_after_concat.weight"], sd["layer_norms_after_concat.bias"], 1e-5)
    h = F.linear(torch.cat([cls, demo], 1), sd["fc_list.0.weight"], sd["fc_list.0.bias"])
    h = F.batch_norm(h, sd["fc_list.1.running_mean"], sd["fc_list.1.running_var"], sd["fc_list.1.weight"],
                     sd["fc_list.1.bias"], training, 0.1, 1e-5)
    return F.linear(torch.relu(h), sd["fc_list.3.weight"], sd["fc_list.3.bias"]).squeeze()   # :178


def bce_with_logits_mean(logits: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """2_train.py:76 + trainer.py:128,176 -- BCEWithLogitsLoss(mean) on output.squeeze()."""
    return F.binary_cross_entropy_with_logits(logits.squeeze(), y.float())


# --------------------------------------------------------------------------
# AdamW + cosine-warmup schedule (2_train.py:110-124)
# --------------------------------------------------------------------------
def adamw_step(p, g, m, v, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8, wd=1e-6):
    """torch.optim.AdamW single-tensor math (decoupled weight decay); in place."""
    p.mul_(1.0 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def cosine_warmup_lr(it: int, first_cycle: int, cycle_mult: float, max_lr: float, min_lr: float,
                     warmup: int, gamma: float) -> float:
    """cosine_annealing_with_warmup_v2.py:67-91 -- lr after ``scheduler.step(it)``."""
    if it >= first_cycle:
        if cycle_mult == 1.0:
            step_in, cycle, cur = it % first_cycle, it // first_cycle, first_cycle
        else:
            n = int(math.log((it / first_cycle * (cycle_mult - 1) + 1), cycle_mult))
            cycle = n
            step_in = it - int(first_cycle * (cycle_mult ** n - 1) / (cycle_mult - 1))
            cur = first_cycle * cycle_mult ** n
    else:
        cur, step_in, cycle = first_cycle, it, 0
    mx = max_lr * (gamma ** cycle)
    if step_in < warmup:
        return (mx - min_lr) * step_in / warmup + min_lr
    return min_lr + (mx - min_lr) * (1 + math.cos(math.pi * (step_in - warmup) / (cur - warmup))) / 2


# --------------------------------------------------------------------------
# one optimisation step -- builder/trainer/trainer.py:20-241 (missing_trainer)
# --------------------------------------------------------------------------
NO_GRAD_PREFIXES = ("img_encoder.",)   # Swin runs under torch.no_grad (tri_mbt_vsltcls.py:208-209)


class OracleTrainer:
    """Holds fp32 parameters (state_dict layout), AdamW moments and the LR
    schedule; ``step`` reproduces trainer.py's train branch, ``evaluate`` its
    test branch.  Parameters that never receive a gradient in the reference
    (Swin; skipped last-layer image/text blocks; unused heads) are left
    untouched exactly as torch.optim.AdamW skips ``grad is None`` tensors."""

    def __init__(self, sd: SD, cfg: Cfg, *, lr_init: float, batch_size: int, iters_per_epoch: int,
                 t_0: int = 50, t_up: int = 5, t_mult: int = 2, gamma: float = 0.5,
                 weight_decay: float = 1e-6, buffers: Sequence[str] = ()):
        self.cfg = cfg
        self.sd = {k: v.clone() for k, v in sd.items()}
        self.param_names = [k for k, v in self.sd.items() if v.is_floating_point() and k not in buffers
                            and not k.endswith(("running_mean", "running_var", ".pe"))]
        self.m = {k: torch.zeros_like(self.sd[k]) for k in self.param_names}
        self.v = {k: torch.zeros_like(self.sd[k]) for k in self.param_names}
        self.steps = {k: 0 for k in self.param_names}
        self.sched = dict(first_cycle=t_0 * iters_per_epoch, cycle_mult=t_mult,
                          max_lr=lr_init * math.sqrt(batch_size), min_lr=1e-6,
                          warmup=t_up * iters_per_epoch, gamma=gamma)
        self.lr = 1e-6            # CosineAnnealingWarmupRestarts.init_lr sets min_lr before the first step
        self.wd = weight_decay
        self.grads: Dict[str, torch.Tensor] = {}

    def _inputs(self, bt):
        """fp16 rounding of x / img_time / txt_time (2_train.py:164, trainer.py:26-27) and the
        ragged trim to the batch max length (trainer.py:41-42)."""
        tmax = int(bt["input_lengths"].max())
        x = bt["x"].half().float()[:, :tmax]
        return x, bt["img_time"].half().float(), bt["txt_time"].half().float()

    def forward(self, bt, training: bool):
        x, img_time, txt_time = self._inputs(bt)
        mnum = missing_to_num(bt["missing"])
        return model_forward(self.sd, self.cfg, x, bt["age"].float(), bt["gen"].float(), bt["input_lengths"],
                             bt["txt"], bt["txt_lengths"], bt["img"], mnum, img_time, txt_time, training)

    def step(self, bt, iteration: int) -> float:
        train = [k for k in self.param_names if not k.startswith(NO_GRAD_PREFIXES)]
        for k in train:
            self.sd[k].requires_grad_(True)
            self.sd[k].grad = None
        logits = self.forward(bt, True)
        # BatchNorm1d running statistics update (momentum 0.1, unbiased var)
        loss = bce_with_logits_mean(logits, bt["y"])
        loss.backward()
        self.grads = {}
        with torch.no_grad():
            for k in train:
                p = self.sd[k]
                if p.grad is None:
                    continue
                self.grads[k] = p.grad.detach().clone()
                self.steps[k] += 1
                adamw_step(p, p.grad, self.m[k], self.v[k], self.steps[k], self.lr, wd=self.wd)
        for k in train:
            self.sd[k].requires_grad_(False)
            self.sd[k].grad = None
        self.lr = cosine_warmup_lr(iteration, **self.sched)         # scheduler.step(iteration), trainer.py:190
        return float(loss)

    def evaluate(self, bt):
        with torch.no_grad():
            logits = self.forward(bt, False)
            loss = bce_with_logits_mean(logits, bt["y"])
        return float(loss), torch.sigmoid(logits.squeeze())
