"""BI_VSLTTXT_MBT_V1 -- MI355X-native drop-in for the reference's two-stream sibling of the tri-modal model
(builder/models/8_missing_models/bi_vslttxt_mbt_v1.py:17-183): vital-sign / lab events (TIE / UMSE
embedding) and clinical text (Linear on BioBERT embeddings + time / modality embedding) fused by
BimodalTransformerEncoder_MBT, CLS vectors mixed per sample by ``missing`` (0: mean of both, 1: vital signs alone), the
same classifier head.  Same constructor, forward signature, return triple and state_dict keys as the reference;
``--input-types vslt_txt`` (the trainer then folds the four modality patterns onto {0, 1}, trainer.py:99-101).

Differences from the reference file, all forced: ``output`` keeps the reference's ``.squeeze()`` ([B]); only
``--vslt-type TIE`` and ``--berttype biobert`` are on the HIP path (the carry-forward / QIE / token-id branches raise).
"""
import torch
import torch.nn as nn

from medical_tri_modal_pilot_amd import ops
from medical_tri_modal_pilot_amd.builder.models.src.transformer.mbt_encoder import BimodalTransformerEncoder_MBT

from .tri_mbt_vsltcls import _compute_dtype, flat_layout


class BI_VSLTTXT_MBT_V1(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.output_dim = args.output_dim
        self.num_layers = args.transformer_num_layers
        self.num_heads = args.transformer_num_head
        self.model_dim = args.transformer_dim
        self.dropout = args.dropout
        self.idx_order = torch.arange(0, args.batch_size).type(torch.LongTensor)
        self.num_nodes = len(args.vitalsign_labtest)
        self.t_len = args.window_size
        self.device = args.device
        self.n_modality = len(args.input_types.split("_"))
        self.bottlenecks_n = 4
        self.compute_dtype = _compute_dtype(args)
        if self.model_dim != 256:
            raise ValueError("transformer_dim must be 256 (Linear(768,256) is hard-coded, bi_vslttxt_mbt_v1.py:78)")
        if args.vslt_type != "TIE" or args.berttype != "biobert":
            raise NotImplementedError("BI_VSLTTXT_MBT_V1 on the MI355X path: --vslt-type TIE and --berttype biobert only")
        self.activations = nn.ModuleDict([
            ["lrelu", nn.LeakyReLU()], ["prelu", nn.PReLU()], ["relu", nn.ReLU(inplace=True)], ["tanh", nn.Tanh()],
            ["sigmoid", nn.Sigmoid()], ["leaky_relu", nn.LeakyReLU(0.2)], ["elu", nn.ELU()]])
        self.relu = self.activations["relu"]

        def embed(n_in):
            return nn.Sequential(nn.Linear(n_in, self.model_dim), nn.LayerNorm(self.model_dim), nn.ReLU(inplace=True))

        self.ie_vslt = embed(1)
        self.ie_time = embed(1)
        self.ie_feat = nn.Embedding(20, self.model_dim)
        self.ie_demo = embed(2)
        self.txt_embedding = nn.Linear(768, self.model_dim)
        self.fusion_transformer = BimodalTransformerEncoder_MBT(
            batch_size=args.batch_size, n_modality=2, bottlenecks_n=4, fusion_startidx=args.mbt_fusion_startIdx,
            d_input=self.model_dim, n_layers=self.num_layers, n_head=self.num_heads, d_model=self.model_dim,
            d_ff=self.model_dim * 4, dropout=self.dropout, txt_idx=1, pe_maxlen=2500, use_pe=[False, True],
            mask=[True, True], compute_dtype=self.compute_dtype)
        classifier_dim = self.model_dim * 2
        self.layer_norms_after_concat = nn.LayerNorm(self.model_dim)
        self.fc_list = nn.Sequential(nn.Linear(classifier_dim, self.model_dim, bias=True), nn.BatchNorm1d(self.model_dim),
                                     self.activations["relu"], nn.Linear(self.model_dim, 1, bias=True))
        if "rmse" in self.args.auxiliary_loss_type:
            self.rmse_layer = nn.Linear(classifier_dim, 1, bias=True)

    def hot_parameters(self):
        """Parameters that receive a gradient on this path, laid out for optim.FlatParams."""
        skip = ("fusion_transformer.layer_norms_after_concat.", "activations.", "rmse_layer.")
        named = [(n, p) for n, p in self.named_parameters() if not n.startswith(skip)]
        return flat_layout(named, self.fusion_transformer.layer_stacks)

    def forward(self, x, h, m, d, x_m, age, gen, input_lengths, txts, txt_lengths, img, missing, f_indices, img_time,
                txt_time, flow_type, reports_tokens, reports_lengths):
        dt = self.compute_dtype
        B = x.size(0)
        age, gen = age.float(), gen.float()
        # vital-sign / lab stream (:139-146) and text stream (:157, 164-165)
        vslt_embedding = ops.TieEmbed.apply(x.float(), self.ie_vslt[0].weight, self.ie_vslt[0].bias, self.ie_vslt[1].weight,
                                            self.ie_vslt[1].bias, self.ie_time[0].weight, self.ie_time[0].bias,
                                            self.ie_time[1].weight, self.ie_time[1].bias, self.ie_feat.weight, dt)
        txt_embedding = ops.DataLinearFn.apply(txts, self.txt_embedding.weight, self.txt_embedding.bias, dt)
        if self.args.imgtxt_time == 1:
            ev = torch.zeros(B, 3, device=x.device)
            ev[:, 0], ev[:, 2] = txt_time.float(), 19.0
            tt = ops.TimeEmbed.apply(ev, self.ie_time[0].weight, self.ie_time[0].bias, self.ie_time[1].weight,
                                     self.ie_time[1].bias, self.ie_feat.weight, dt)
            txt_embedding = txt_embedding + tt.unsqueeze(1)
        missing = missing.to(x.device).long()
        outputs, _ = self.fusion_transformer(
            enc_outputs=[vslt_embedding, txt_embedding], fixed_lengths=[vslt_embedding.size(1), txt_embedding.size(1)],
            varying_lengths=[input_lengths, txt_lengths + 2], fusion_idx=None, missing=missing)
        return self._head(outputs, age, gen, missing, B)

    def _head(self, outputs, age, gen, missing, B):
        """The classifier on the two CLS rows -- a hook: sibling models mix them differently."""
        # per-sample CLS mix (:171-174): mean of both streams' CLS rows, or the vital-sign CLS alone
        c0, c1 = outputs[0][:, 0, :].float(), outputs[1][:, 0, :].float()
        cls = torch.where((missing == 0).unsqueeze(1), torch.stack([c0, c1]).mean(0), c0)
        fused_head = B <= ops.HEAD_MAX_B and "rmse" not in self.args.auxiliary_loss_type and (B > 1 or not self.training)
        if fused_head:                   # LN(cls) | ie_demo(age, gender) -> fc_list as six HIP launches (ops.HeadFn)
            bn, ln, dm = self.fc_list[1], self.layer_norms_after_concat, self.ie_demo
            nbt = bn.num_batches_tracked if (bn.training and bn.track_running_stats) else None
            if nbt is not None and not nbt.is_cuda:
                nbt.add_(1)
                nbt = None
            use_batch = bn.training or not bn.track_running_stats
            out = ops.HeadFn.apply(cls, age, gen, use_batch, 0.1 if bn.momentum is None else bn.momentum, bn.eps,
                                   bn.running_mean, bn.running_var, nbt, dm[0].weight, dm[0].bias, dm[1].weight, dm[1].bias,
                                   ln.weight, ln.bias, self.fc_list[0].weight, self.fc_list[0].bias, bn.weight, bn.bias,
                                   self.fc_list[3].weight, self.fc_list[3].bias)
            return out.squeeze(), None, None
        demo_embedding = self.ie_demo(torch.stack([age, gen], dim=1))
        class_input = torch.cat([self.layer_norms_after_concat(cls), demo_embedding], dim=1)
        output2 = self.rmse_layer(class_input).squeeze() if "rmse" in self.args.auxiliary_loss_type else None
        return self.fc_list(class_input).squeeze(), output2, None
