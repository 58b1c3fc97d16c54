"""TRI_MBT_V1 -- MI355X-native drop-in for the reference's first tri-modal MBT model
(builder/models/8_missing_models/tri_mbt_v1.py:17-283).

Same embeddings, same fusion encoder kernels, same forward signature and state_dict keys as the reference class; it
differs from TRI_MBT_VSLTCLS in what it reads of the encoder and in its head:

  * the encoder is built WITHOUT ``vsltonly`` (reference :129-145): every layer runs all three streams, and the CLS rows of
    ALL THREE streams are read (:271) -- so nothing of the last layer is dead here, and the vital-sign stream keeps the padded
    layout (ops.FusionStackFn hands out the three [B, N, 256] results);
  * head (:271-281): LayerNorm over the three CLS rows, the demographic embedding appended to each, ``fc_list`` = Linear ->
    **LayerNorm** -> ReLU -> Linear (no BatchNorm, :154-159) applied to the three rows, then per sample the mean of the logits
    of the modalities that are present: candidates (all three, vslt + image, vslt + text, vslt) gathered by ``missing``.

Samples without an image still skip the frozen encoder (--skip-missing-images): their image CLS logit is not among the
candidates their ``missing`` id selects, and ``fc_list`` has no batch statistics, so it feeds nothing.
"""
import torch
import torch.nn as nn

from .tri_mbt_vsltcls import TRI_MBT_VSLTCLS, flat_layout


class TRI_MBT_V1(TRI_MBT_VSLTCLS):
    head_fusable = False          # three CLS rows through a LayerNorm head: the torch form below

    def __init__(self, args):
        super().__init__(args)
        enc = self.fusion_transformer
        enc.vsltonly = 0                              # (:129-145: the encoder's default)
        enc.first_stream_output_only = False
        enc.pack_rows = False
        classifier_dim = self.model_dim if self.args.vslt_type == "QIE" else self.model_dim * 2
        # same attribute, same position in the module order: Linear, LayerNorm, ReLU, Linear (:154-159)
        self.fc_list = nn.Sequential(nn.Linear(classifier_dim, self.model_dim, bias=True), nn.LayerNorm(self.model_dim),
                                     self.activations["relu"], nn.Linear(self.model_dim, self.output_dim, bias=True))

    def hot_parameters(self):
        skip = ["img_encoder.", "fusion_transformer.layer_norms_after_concat.", "activations.", "rmse_layer."]
        named = [(n, p) for n, p in self.named_parameters() if not n.startswith(tuple(skip))]
        return flat_layout(named, self.fusion_transformer.layer_stacks)

    def _head(self, outputs, demo_embedding, age, gen, missing, fused_head):
        stack = torch.stack([outputs[0][:, 0, :], outputs[1][:, 0, :], outputs[2][:, 0, :]]).float()     # vslt, img, txt
        stack = self.layer_norms_after_concat(stack)
        if self.args.vslt_type != "QIE":
            stack = torch.cat([stack, demo_embedding.unsqueeze(0).expand(3, -1, -1)], dim=2)
        o = self.fc_list(stack).squeeze(-1)                                                                  # [3, B]
        cands = torch.stack([o.mean(0), torch.stack([o[0], o[1]]).mean(0), torch.stack([o[0], o[2]]).mean(0), o[0]])
        idx = torch.arange(o.shape[1], device=o.device)
        return cands[missing.to(o.device).long(), idx], None, None
