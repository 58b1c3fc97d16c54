"""TRI_MBT_V2 -- MI355X-native drop-in for the reference's second tri-modal MBT model
(builder/models/8_missing_models/tri_mbt_v2.py:17-262; SURVEY 8 f-4, VERDICT r4 missing #3).

TRI_MBT_V1's structure -- all three streams through every layer, the three CLS rows read -- with these differences, all taken
from the reference class:

  * the Swin-T image encoder is TRAINED: ``self.img_encoder(img)`` runs with gradients (:208-211, no ``torch.no_grad()``, no
    ``.eval()`` in the constructor), so its 171 parameter tensors are part of ``hot_parameters()`` and the forward takes the
    encoder's autograd path (SwinTransformer.forward_train: mtmp_gemm_nt / mtmp_gemm_tn, mtmp_layernorm_rows(_bwd),
    mtmp_gelu_fwd / _bwd, mtmp_swin_window_attn(_bwd));
  * the reports arrive as TOKEN IDS: ``self.txt_embedding(txts.type(torch.LongTensor))`` (:205) only works on the
    ``nn.Embedding(30000, 256)`` of ``--berttype bert`` (:79-80) -- with the default ``biobert`` Linear the reference raises, and
    so does this class, at construction;
  * head (:234-256): LayerNorm over the three CLS rows, flattened to [3 B, 256], the demographic embedding appended, ``fc_list`` =
    Linear -> **BatchNorm1d over the 3 B rows** -> ReLU -> Linear(``output_dim``), the per-sample mean over the present modalities
    gathered by ``missing``; returns [B, output_dim] (no squeeze); ``rmse_layer`` exists only with an rmse auxiliary loss (:168-169);
  * one image per sample, image keys unmasked (``mask=[True, False, True]``, :133); no ``--residual-bottlenecks`` (:118-134).
"""
import torch
import torch.nn as nn

from .tri_mbt_v1 import TRI_MBT_V1
from .tri_mbt_vsltcls import flat_layout


class TRI_MBT_V2(TRI_MBT_V1):
    TRAINS_ENCODER_IN_REFERENCE = True

    def __init__(self, args):
        if args.berttype != "bert":
            raise NotImplementedError("TRI_MBT_V2 feeds txts.type(torch.LongTensor) to txt_embedding (tri_mbt_v2.py:205): token ids, "
                                      "--berttype bert")
        if int(getattr(args, "multiimages", 0)) == 1 or int(getattr(args, "residual_bottlenecks", 0)) == 1:
            raise NotImplementedError("TRI_MBT_V2: one image per sample and no residual bottlenecks (tri_mbt_v2.py:118-134)")
        super().__init__(args)
        self.output_dim = args.output_dim                                                   # (:26)
        self.img_encoder.train()                                                            # (no .eval() in the reference's constructor)
        if "rmse" not in self.args.auxiliary_loss_type and "rmse_layer" in self._modules:  # (:168-169)
            del self._modules["rmse_layer"]
        classifier_dim = self.model_dim if self.args.vslt_type == "QIE" else self.model_dim * 2
        self.fc_list = nn.Sequential(nn.Linear(classifier_dim, self.model_dim, bias=True), nn.BatchNorm1d(self.model_dim),
                                     self.activations["relu"], nn.Linear(self.model_dim, self.output_dim, bias=True))

    def forward(self, x, h, m, d, x_m, age, gen, input_lengths, txts, txt_lengths, img, missing, f_indices, img_time, txt_time,
                flow_type, reports_tokens, reports_lengths):
        # (:205) the reports are token ids whatever dtype the loader hands them over in
        return super().forward(x, h, m, d, x_m, age, gen, input_lengths, txts.long(), txt_lengths, img, missing, f_indices, img_time,
                               txt_time, flow_type, reports_tokens, reports_lengths)

    def hot_parameters(self):
        skip = ["img_encoder.head.", "fusion_transformer.layer_norms_after_concat.", "activations.", "rmse_layer."]
        named = [(n, p) for n, p in self.named_parameters() if not n.startswith(tuple(skip))]
        return flat_layout(named, self.fusion_transformer.layer_stacks)

    def _head(self, outputs, demo_embedding, age, gen, missing, fused_head):
        stack = torch.stack([outputs[0][:, 0, :], outputs[1][:, 0, :], outputs[2][:, 0, :]]).float()     # vslt, img, txt
        class_input = self.layer_norms_after_concat(stack).reshape(-1, self.model_dim)                       # [3 B, 256] (:234-235)
        if self.args.vslt_type != "QIE":
            class_input = torch.cat([class_input, demo_embedding.repeat(3, 1)], dim=1)
        o = self.fc_list(class_input).reshape(3, -1, self.output_dim)                                         # [3, B, out]
        cands = torch.stack([o.mean(0), torch.stack([o[0], o[1]]).mean(0), torch.stack([o[0], o[2]]).mean(0), o[0]])
        idx = torch.arange(o.shape[1], device=o.device)
        output2 = None
        if "rmse" in self.args.auxiliary_loss_type:
            r = self.rmse_layer(class_input).reshape(3, -1)
            rc = torch.stack([r.mean(0), torch.stack([r[0], r[1]]).mean(0), torch.stack([r[0], r[2]]).mean(0), r[0]])
            output2 = rc[missing.to(o.device).long(), idx]
        return cands[missing.to(o.device).long(), idx], output2, None
