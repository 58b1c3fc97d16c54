"""BITXT_MBT_VFLEXIBLE1 -- MI355X-native drop-in for the reference's two-stream (vital signs + text) MBT model with LEARNED
modality weights (builder/models/8_missing_models/bitxt_mbt_vflexible1.py:17-200).

BI_VSLTTXT_MBT_V1's embeddings and BimodalTransformerEncoder_MBT, with the head of the "flexible" family: LayerNorm over the
two CLS rows, the demographic embedding appended to each, ``fc_list`` = Linear -> LayerNorm -> ReLU -> Linear applied to both
(:177-182), the two logits weighted by softmax(flexibleavg) over the PRESENT modalities (an absent text stream is filled
with -1e9 first, :184-188), summed, and gathered per sample by ``missing`` (0: both, 1: vital signs alone, :190-192).
``rmse_layer`` is always built (:117).  The reference makes its mask table with ``.cuda()`` in ``__init__``; here it is a
constant moved to the logits' device.
"""
import torch
import torch.nn as nn

from .bi_vslttxt_mbt_v1 import BI_VSLTTXT_MBT_V1, flat_layout

_ABSENT = torch.tensor([[False, False], [False, True]])          # rows = missing (0: both, 1: vslt only), columns = (vslt, text)


class BITXT_MBT_VFLEXIBLE1(BI_VSLTTXT_MBT_V1):
    def __init__(self, args):
        super().__init__(args)
        self.output_dim = 1                                       # (:26)
        self.flexibleavg = nn.Parameter(torch.zeros(2, 1))        # (:102) a root-level parameter: first in parameters() here and there
        classifier_dim = self.model_dim * 2
        # the reference's order of the classifier modules (:117-124): rmse_layer, layer_norms_after_concat, fc_list
        for name in ("layer_norms_after_concat", "fc_list", "rmse_layer", "relu"):
            if name in self._modules:
                del self._modules[name]
        self.rmse_layer = nn.Linear(classifier_dim, 1, bias=True)
        self.layer_norms_after_concat = nn.LayerNorm(self.model_dim)
        self.fc_list = nn.Sequential(nn.Linear(classifier_dim, self.model_dim, bias=True), nn.LayerNorm(self.model_dim),
                                     self.activations["relu"], nn.Linear(self.model_dim, self.output_dim, bias=True))

    def hot_parameters(self):
        skip = ("fusion_transformer.layer_norms_after_concat.", "activations.", "rmse_layer.")
        named = [(n, p) for n, p in self.named_parameters() if not n.startswith(skip)]
        return flat_layout(named, self.fusion_transformer.layer_stacks)

    def _head(self, outputs, age, gen, missing, B):
        stack = torch.stack([outputs[0][:, 0, :], outputs[1][:, 0, :]]).float()                # [2, B, 256]
        stack = self.layer_norms_after_concat(stack)
        demo = self.ie_demo(torch.stack([age, gen], dim=1))
        o = self.fc_list(torch.cat([stack, demo.unsqueeze(0).expand(2, -1, -1)], dim=2))        # [2, B, 1]
        miss = missing.to(o.device).long()
        w = self.flexibleavg.float().repeat(1, B).masked_fill(_ABSENT.to(o.device)[miss].permute(1, 0), -1e9)
        o = o * torch.softmax(w, dim=0).unsqueeze(2)
        cands = torch.stack([o[0] + o[1], o[0]])
        return cands[miss, torch.arange(B, device=o.device)], None, None
