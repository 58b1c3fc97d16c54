"""TRI_MBT_VFLEXIBLE -- MI355X-native drop-in for the reference's tri-modal MBT model with LEARNED modality weights
(builder/models/8_missing_models/tri_mbt_vflexible.py:17-297).

TRI_MBT_V1 (all three CLS rows through the LayerNorm head) with one more parameter, ``flexibleavg`` [3, 1] (:148): instead of
the plain mean over the modalities that are present, the three logits are weighted by softmax(flexibleavg * temperature) taken
over the PRESENT modalities only (absent ones are filled with -1e9 first, :276-279: their weight is exactly 0), and the
weighted logits are summed (:282-286).  ``tri_mbt_vflexible2`` / ``3`` are the same model with temperature 10 / 3.334.

The reference builds its mask table with ``.cuda()`` inside ``__init__`` (:150-165); here it is a registered-free constant made
on the logits' device.
"""
import torch
import torch.nn as nn

from .tri_mbt_v1 import TRI_MBT_V1

# rows = missing_num (0: all three, 1: vslt + image, 2: vslt + text, 3: vslt), columns = (vslt, image, text): True = absent
_ABSENT = torch.tensor([[False, False, False], [False, False, True], [False, True, False], [False, True, True]])


class TRI_MBT_VFLEXIBLE(TRI_MBT_V1):
    flex_temperature = 1.0

    def __init__(self, args):
        super().__init__(args)
        self.flexibleavg = nn.Parameter(torch.zeros(3, 1))          # (:148; registered behind fc_list as in the reference)

    def hot_parameters(self):
        return super().hot_parameters()                             # flexibleavg is trained: nothing more to skip

    def _head(self, outputs, demo_embedding, age, gen, missing, fused_head):
        stack = torch.stack([outputs[0][:, 0, :], outputs[1][:, 0, :], outputs[2][:, 0, :]]).float()     # vslt, img, txt
        stack = self.layer_norms_after_concat(stack)
        if self.args.vslt_type != "QIE":
            stack = torch.cat([stack, demo_embedding.unsqueeze(0).expand(3, -1, -1)], dim=2)
        o = self.fc_list(stack)                                                                              # [3, B, 1]
        B = o.shape[1]
        miss = missing.to(o.device).long()
        w = self.flexibleavg.float().repeat(1, B)                                                            # [3, B]
        w = w.masked_fill(_ABSENT.to(o.device)[miss].permute(1, 0), -1e9)
        o = o * torch.softmax(w * self.flex_temperature, dim=0).unsqueeze(2)
        cands = torch.stack([o.sum(0), o[0] + o[1], o[0] + o[2], o[0]])                                      # [4, B, 1]
        return cands[miss, torch.arange(B, device=o.device)], None, None
