"""Pins oracle/tri_mbt_oracle.py against golden vectors produced by the REAL
reference (tests/golden/gen/make_golden.py, run in the build container).
CPU only; the oracle is test infrastructure."""
import math
import os

import numpy as np
import pytest
import torch

import filler
from oracle import tri_mbt_oracle as O

ROWSTEP = 5
TOL = dict(rtol=2e-5, atol=2e-6)


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def _filled(keys_shapes, prefix):
    return {k: filler.fill_tensor(prefix + k, torch.zeros(s)) for k, s in keys_shapes.items()}


MHA_KEYS = {f"{n}_proj.linear.{w}": s for n in ("query", "key", "value")
            for w, s in (("weight", (256, 256)), ("bias", (256,)))}


def digest(t):
    f = t.detach().reshape(-1).double()
    idx = torch.linspace(0, f.numel() - 1, 8).long()
    return torch.cat([f.norm().view(1), f[idx]]).numpy()


def test_mha_blocks(golden_dir):
    G = _g(golden_dir, "blocks")
    sd = {"a." + k: v for k, v in _filled(MHA_KEYS, "g1.").items()}
    g = torch.Generator().manual_seed(11)
    for N in (54, 133, 261):
        x = torch.randn(4, N, 256, generator=g, requires_grad=True)
        lens = torch.from_numpy(G[f"mha{N}_len"])
        for k in sd:
            sd[k].requires_grad_(True)
            sd[k].grad = None
        y = O.multi_head_attention(sd, "a", x, O.key_pad_mask(N, lens), 4)
        w = torch.randn(y.shape, generator=g)
        (y * w).sum().backward()
        np.testing.assert_allclose(y[:, ::ROWSTEP].detach().numpy(), G[f"mha{N}_y"], **TOL)
        np.testing.assert_allclose(x.grad[:, ::ROWSTEP].numpy(), G[f"mha{N}_dx"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(digest(sd["a.query_proj.linear.weight"].grad), G[f"mha{N}_dWq"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(sd["a.value_proj.linear.bias"].grad.numpy(), G[f"mha{N}_dbv"], rtol=1e-4, atol=1e-4)
    for k in sd:
        sd[k].requires_grad_(False)
    x = torch.randn(2, 40, 256, generator=g)
    y = O.multi_head_attention(sd, "a", x, O.key_pad_mask(40, torch.tensor([0, 17])), 4)
    np.testing.assert_allclose(y[:, ::ROWSTEP].numpy(), G["mha_full_y"], **TOL)
    # a fully masked row is the plain mean of V over all keys (SURVEY §7 invariant iii)
    v = torch.nn.functional.linear(x[0], sd["a.value_proj.linear.weight"], sd["a.value_proj.linear.bias"])
    np.testing.assert_allclose(y[0, 3].numpy(), v.mean(0).numpy(), rtol=1e-5, atol=1e-6)
    y = O.multi_head_attention(sd, "a", x, None, 4)
    np.testing.assert_allclose(y[:, ::ROWSTEP].numpy(), G["mha_nomask_y"], **TOL)


def test_custom_layernorm_and_layer(golden_dir):
    G = _g(golden_dir, "blocks")
    g = torch.Generator().manual_seed(11)
    # consume the generator exactly as the golden script did
    for N in (54, 133, 261):
        torch.randn(4, N, 256, generator=g)
        torch.randn(4, N, 256, generator=g)
    torch.randn(2, 40, 256, generator=g)
    ln = _filled({"gamma": (256,), "beta": (256,)}, "g2.")
    z = (torch.randn(3, 17, 256, generator=g) * 2 + 0.3).requires_grad_()
    ga, be = ln["gamma"].requires_grad_(), ln["beta"].requires_grad_()
    y = O.custom_layernorm(z, ga, be)
    w = torch.randn(y.shape, generator=g)
    (y * w).sum().backward()
    np.testing.assert_allclose(y.detach().numpy(), G["ln_y"], **TOL)
    np.testing.assert_allclose(z.grad.numpy(), G["ln_dz"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ga.grad.numpy(), G["ln_dgamma"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(be.grad.numpy(), G["ln_dbeta"], rtol=1e-4, atol=1e-4)
    keys = {"attention_prenorm.gamma": (256,), "attention_prenorm.beta": (256,),
            "feed_forward_prenorm.gamma": (256,), "feed_forward_prenorm.beta": (256,),
            "feed_forward.w_1.weight": (1024, 256, 1), "feed_forward.w_1.bias": (1024,),
            "feed_forward.w_2.weight": (256, 1024, 1), "feed_forward.w_2.bias": (256,)}
    keys.update({"self_attention." + k: s for k, s in MHA_KEYS.items()})
    sd = {"L." + k: v.requires_grad_() for k, v in _filled(keys, "g3.").items()}
    x = torch.randn(3, 70, 256, generator=g, requires_grad=True)
    lens = torch.from_numpy(G["lay_len"])
    y = O.encoder_layer(sd, "L", x, O.key_pad_mask(70, lens), 4)
    w = torch.randn(y.shape, generator=g)
    (y * w).sum().backward()
    np.testing.assert_allclose(y[:, ::ROWSTEP].detach().numpy(), G["lay_y"], rtol=5e-5, atol=5e-6)
    np.testing.assert_allclose(x.grad[:, ::ROWSTEP].numpy(), G["lay_dx"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(digest(sd["L.feed_forward.w_1.weight"].grad), G["lay_dW1"], rtol=2e-4, atol=1e-4)
    np.testing.assert_allclose(digest(sd["L.self_attention.key_proj.linear.weight"].grad), G["lay_dWk"], rtol=2e-4, atol=1e-4)
    np.testing.assert_allclose(sd["L.attention_prenorm.gamma"].grad.numpy(), G["lay_dgamma_attn"], rtol=2e-4, atol=2e-4)


def encoder_keys(L, n_mod=3):
    keys = {"layer_norms_after_concat.weight": (256,), "layer_norms_after_concat.bias": (256,),
            "bottlenecks": (1, 4, 256), "positional_encoding.pe": (1, 2500, 256)}
    for m in range(n_mod):
        keys[f"cls_token_per_modality.{m}"] = (1, 1, 256)
        keys[f"layer_norms_in.{m}.weight"] = (256,)
        keys[f"layer_norms_in.{m}.bias"] = (256,)
    for l in range(L):
        for m in range(n_mod):
            p = f"layer_stacks.{l}.{m}."
            for n in ("attention_prenorm", "feed_forward_prenorm"):
                keys[p + n + ".gamma"] = (256,)
                keys[p + n + ".beta"] = (256,)
            for k, s in MHA_KEYS.items():
                keys[p + "self_attention." + k] = s
            keys[p + "feed_forward.w_1.weight"] = (1024, 256, 1)
            keys[p + "feed_forward.w_1.bias"] = (1024,)
            keys[p + "feed_forward.w_2.weight"] = (256, 1024, 1)
            keys[p + "feed_forward.w_2.bias"] = (256,)
    return keys


def test_mbt_encoder_all_variants(golden_dir):
    G = _g(golden_dir, "encoder")
    n = int(G["n_cases"])
    assert n == 16
    for case in range(n):
        vsltonly, resb, fstart, multi, B, T, L = [int(v) for v in G[f"c{case}_cfg"]]
        sd = {"f." + k: v for k, v in _filled(encoder_keys(L), "g4.").items()}
        g = torch.Generator().manual_seed(100 + case)
        n_img = 147 if multi else 49
        v = torch.randn(B, T, 256, generator=g)
        i = torch.randn(B, n_img, 256, generator=g)
        t = torch.randn(B, 30, 256, generator=g)
        in_len, txt_len = torch.tensor([T, 3, 11, 7]), torch.tensor([20, 0, 5, 0])
        img_time = None
        if multi:                      # rebuild an img_time whose "!= 10" count equals the golden's
            cnt = torch.from_numpy(G[f"c{case}_imgcnt"])
            img_time = torch.full((B, 3), 10.0)
            for b in range(B):
                img_time[b, : int(cnt[b])] = -1.0
        outs, _ = O.mbt_encoder(sd, "f", [v, i, t], in_len, txt_len, img_time, torch.tensor([0, 1, 2, 3]),
                                n_layers=L, n_head=4, fusion_startidx=fstart, vsltonly=vsltonly,
                                resbottle=bool(resb), multiimages=multi)
        assert len(outs) == (1 if vsltonly else 3)
        for m, o in enumerate(outs):
            ref = G[f"c{case}_out{m}"]
            got = o if m == 0 else o[:, ::7]
            np.testing.assert_allclose(got.numpy(), ref, rtol=1e-4, atol=2e-5, err_msg=f"case {case} stream {m}")


def test_bimodal_mbt_encoder(golden_dir):
    """the restated BimodalTransformerEncoder_MBT against outputs and gradients of the real class."""
    G = _g(golden_dir, "bimodal")
    n = int(G["n_cases"])
    assert n == 4
    for case in range(n):
        mask1, txt_idx, pe1, B, T, L = [int(v) for v in G[f"c{case}_cfg"]]
        sd = {"f." + k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith(".pe"))
              for k, v in _filled(encoder_keys(L, 2), "g5.").items()}
        g = torch.Generator().manual_seed(300 + case)
        v = torch.randn(B, T, 256, generator=g).requires_grad_()
        t = torch.randn(B, 30, 256, generator=g).requires_grad_()
        in_len, txt_len = torch.tensor([T, 3, 11, 7]), torch.tensor([20, 0, 5, 0])
        outs, _ = O.mbt_encoder_bimodal(sd, "f", [v, t], [in_len, txt_len + 2], torch.tensor([0, 1, 1, 0]), n_layers=L,
                                        n_head=4, txt_idx=txt_idx, use_pe=(False, bool(pe1)), mask=(True, bool(mask1)))
        np.testing.assert_allclose(outs[0].detach().numpy(), G[f"c{case}_out0"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(outs[1][:, ::7].detach().numpy(), G[f"c{case}_out1"], rtol=1e-4, atol=2e-5)
        w0 = torch.randn(outs[0].shape, generator=g)
        w1 = torch.randn(outs[1].shape, generator=g)
        ((outs[0] * w0).sum() + (outs[1] * w1).sum()).backward()
        np.testing.assert_allclose(v.grad[:, ::ROWSTEP].numpy(), G[f"c{case}_dv"], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(t.grad[:, ::ROWSTEP].numpy(), G[f"c{case}_dt"], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(sd["f.bottlenecks"].grad.numpy(), G[f"c{case}_dbott"], rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(digest(sd["f.layer_stacks.1.1.feed_forward.w_1.weight"].grad), G[f"c{case}_dw1"], rtol=2e-4, atol=1e-4)
        np.testing.assert_allclose(digest(sd["f.layer_stacks.0.0.self_attention.query_proj.linear.weight"].grad), G[f"c{case}_dwq"],
                                   rtol=2e-4, atol=1e-4)


def _bimodel_sd(golden_dir):
    import json
    with open(os.path.join(golden_dir, "state_shapes_bi_vslttxt_L2.json")) as f:
        shapes = json.load(f)
    sd = {k: filler.fill_tensor(k, torch.zeros(s)) for k, (s, dt) in shapes.items() if dt.startswith("float")}
    sd["fusion_transformer.positional_encoding.pe"] = O.sinusoid_table(2500, 256).unsqueeze(0)
    return sd


def test_bi_vslttxt_model_step(golden_dir):
    """the restated BI_VSLTTXT_MBT_V1 against logits, loss and all 86 parameter gradients of the real class."""
    G = _g(golden_dir, "bimodel_step")
    sd = _bimodel_sd(golden_dir)
    names = [str(s) for s in G["grad_names"]]
    for k in names:
        sd[k].requires_grad_(True)
    bt = filler.make_batch(int(G["seed"]), int(G["B"]), int(G["T"]))
    mnum = torch.from_numpy(G["missing_num"])
    tmax = int(bt["input_lengths"].max())
    out = O.bi_vslttxt_forward(sd, bt["x"][:, :tmax], bt["age"], bt["gen"], bt["input_lengths"], bt["txt"], bt["txt_lengths"],
                               mnum, bt["txt_time"].half().float(), n_layers=2, training=True)
    np.testing.assert_allclose(out.detach().numpy(), G["logits"], rtol=1e-4, atol=1e-5)
    loss = O.bce_with_logits_mean(out, bt["y"])
    assert abs(float(loss) - float(G["loss"])) < 1e-5
    loss.backward()
    assert all(sd[str(k)].grad is None for k in G["nograd_names"] if str(k) in sd)
    for n_, gd in zip(names, G["grad_digest"]):
        np.testing.assert_allclose(digest(sd[n_].grad), gd, rtol=1e-4, atol=2e-6, err_msg="grad " + n_)


def test_missing_num_and_lengths(golden_dir):
    G = _g(golden_dir, "model_step")
    bt = filler.make_batch(int(G["seed"]), int(G["B"]), int(G["T"]))
    mn = O.missing_to_num(bt["missing"])
    assert torch.equal(mn, torch.from_numpy(G["missing_num"]))
    assert torch.equal(mn, bt["missing_num"])
    # every pattern, and a batch that lacks some patterns
    pat = torch.tensor([[0., 1., 1.], [0., 0., 0.], [0., 0., 0.], [0., 1., 0.]])
    assert O.missing_to_num(pat).tolist() == [3, 0, 0, 2]
    # the reference mutates input_lengths in place (+1, mbt_encoder.py:704)
    assert torch.equal(bt["input_lengths"] + 1, torch.from_numpy(G["input_lengths_after"]))
    kv = O.fusion_kv_lengths(torch.tensor([5, 1000]), torch.tensor([0, 126]), None, 0, 49)
    assert kv[0].tolist() == [10, 1005] and kv[1] is None and kv[2].tolist() == [4, 133]


def test_scheduler_lr_sequence(golden_dir):
    G = _g(golden_dir, "sched")
    for it, lr in zip(G["its"], G["lrs"]):
        got = O.cosine_warmup_lr(int(it), 500, 2, 1e-5 * math.sqrt(64), 1e-6, 50, 0.5)
        assert abs(got - lr) <= 1e-12 + 1e-9 * abs(lr), (it, got, lr)
    # SURVEY.md §8c known answers
    assert abs(O.cosine_warmup_lr(50, 500, 2, 8e-5, 1e-6, 50, 0.5) - 8e-5) < 1e-12
    assert abs(O.cosine_warmup_lr(1, 500, 2, 8e-5, 1e-6, 50, 0.5) - 2.58e-6) < 1e-9


def model_state_shapes(L, multi):
    """Key -> shape of the reference module's state_dict (floating tensors), built from the product
    module definition's parameter census would be circular; spelled out here instead."""
    from tests.state_shapes import reference_state_shapes
    return reference_state_shapes(L)


def _model_sd(L):
    shapes = model_state_shapes(L, 0)
    sd = {k: filler.fill_tensor(k, torch.zeros(s)) for k, s in shapes.items()}
    sd["fusion_transformer.positional_encoding.pe"] = O.sinusoid_table(2500, 256).unsqueeze(0)
    return sd


def test_swin_forward(golden_dir):
    G = _g(golden_dir, "swin")
    sd = _model_sd(2)
    g = torch.Generator().manual_seed(int(G["seed"]))
    img = torch.rand(2, 1, 224, 224, generator=g)
    with torch.no_grad():
        y = O.swin_forward(sd, "img_encoder", img)
    np.testing.assert_allclose(y.numpy(), G["feat"], rtol=2e-4, atol=2e-5)


def _swin_train_pairs(G):
    """the 12 (attention-branch, MLP-branch) row-scale pairs of O.swin_forward from the golden's recorded StochasticDepth draws
    (call order; blocks with p = 0 made no draw)"""
    draws, it = torch.from_numpy(G["draws"]), 0
    pairs = []
    for p_ in G["p"]:
        if p_ == 0.0:
            pairs.append((None, None))
        else:
            pairs.append((draws[it], draws[it + 1]))
            it += 2
    assert it == draws.shape[0]
    return pairs


def test_swin_forward_train_mode_stochastic_depth(golden_dir):
    """The real encoder in TRAIN mode (tests/golden/gen/make_golden.py gen_swin_train): the reference's SwinTransformerBlock decides
    where the row-mode StochasticDepth noise multiplies (swin_transformer.py:437,448-449); the draws it used are in the fixture.
    Pins the oracle's ``row_scales`` path -- what the HIP encoder's train mode is checked against."""
    G = _g(golden_dir, "swin_train")
    sd = _model_sd(2)
    g = torch.Generator().manual_seed(int(G["seed"]))
    img = torch.rand(3, 1, 224, 224, generator=g)
    assert float(G["draws"].min()) == 0.0                         # a dropped branch is part of the case
    with torch.no_grad():
        y = O.swin_forward(sd, "img_encoder", img, row_scales=_swin_train_pairs(G))
        y_eval = O.swin_forward(sd, "img_encoder", img)
    np.testing.assert_allclose(y.numpy(), G["feat"], rtol=2e-4, atol=2e-5)
    assert float(np.abs(y_eval.numpy() - G["feat"]).max()) > 1e-2  # (the noise matters: eval mode is far from it)


@pytest.mark.parametrize("size", [512, 200])
def test_swin_forward_padded_windows_and_odd_merges(golden_dir, size):
    """--image-size 512 (maps of 128 / 64 / 32 / 16 tokens a side: every stage zero-pads its windows, swin_transformer.py:150-152)
    and a 200 x 200 input (50 -> 25 -> 13 -> 7: two odd-sized patch merges, :34-44): the oracle against the real class."""
    G = _g(golden_dir, "swin_sizes")
    sd = _model_sd(2)
    g = torch.Generator().manual_seed(int(G[f"seed{size}"]))
    img = torch.rand(1 if size == 512 else 2, 1, size, size, generator=g)
    with torch.no_grad():
        y = O.swin_forward(sd, "img_encoder", img)
    np.testing.assert_allclose(y.numpy(), G[f"feat{size}"], rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("multi,tag", [(0, "model_step"), (1, "model_step_multi")])
def test_full_training_step(golden_dir, multi, tag):
    G = _g(golden_dir, tag)
    sd = _model_sd(2)
    cfg = O.Cfg(n_layers=2, multiimages=multi)
    bt = filler.make_batch(int(G["seed"]), int(G["B"]), int(G["T"]), multiimages=multi)
    tr = O.OracleTrainer(sd, cfg, lr_init=1e-5, batch_size=4, iters_per_epoch=10)
    if multi == 0:
        x, _, _ = tr._inputs(bt)
        psd = {k: v.clone().requires_grad_() for k, v in sd.items() if k.startswith("ie_")}
        emb = O.tie_embedding(psd, bt["x"])
        np.testing.assert_allclose(emb.detach().numpy(), G["tie_emb"], rtol=2e-5, atol=2e-6)
        (emb * torch.from_numpy(G["tie_w"])).sum().backward()
        np.testing.assert_allclose(psd["ie_vslt.0.weight"].grad.numpy(), G["tie_dWv"], rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(psd["ie_time.0.bias"].grad.numpy(), G["tie_dbt"], rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(psd["ie_vslt.1.weight"].grad.numpy(), G["tie_dgv"], rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(psd["ie_feat.weight"].grad.numpy(), G["tie_dF"], rtol=2e-4, atol=2e-4)
    _, sig0 = tr.evaluate(bt)
    ev = torch.sigmoid(torch.from_numpy(G["eval_logits"]).squeeze())
    np.testing.assert_allclose(sig0.numpy(), ev.numpy(), rtol=1e-4, atol=1e-6)
    loss = tr.step(bt, 1)
    assert abs(loss - float(G["loss"])) < 1e-5 * max(1.0, abs(float(G["loss"])))
    assert abs(tr.lr - float(G["lr_after"])) < 1e-12
    names = [str(s) for s in G["grad_names"]]
    assert sorted(names) == sorted(tr.grads.keys())
    nograd = set(str(s) for s in G["nograd_names"])
    assert nograd.isdisjoint(tr.grads.keys())
    for n_, gd, pd in zip(names, G["grad_digest"], G["param_digest"]):
        np.testing.assert_allclose(digest(tr.grads[n_]), gd, rtol=1e-4, atol=2e-6, err_msg="grad " + n_)
        np.testing.assert_allclose(digest(tr.sd[n_]), pd, rtol=1e-5, atol=1e-7, err_msg="param " + n_)
    if "loss2" in G.files:
        loss2 = tr.step(bt, 2)
        assert abs(loss2 - float(G["loss2"])) < 1e-5
        # the golden's BatchNorm buffers were captured after the second train step
        np.testing.assert_allclose(tr.sd["fc_list.1.running_mean"].numpy(), G["bn_running_mean"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(tr.sd["fc_list.1.running_var"].numpy(), G["bn_running_var"], rtol=1e-4, atol=1e-6)
        tl, sg = tr.evaluate(bt)
        assert abs(tl - float(G["test_loss"])) < 1e-5
        np.testing.assert_allclose(sg.numpy(), G["test_sigmoid"], rtol=1e-4, atol=1e-6)
