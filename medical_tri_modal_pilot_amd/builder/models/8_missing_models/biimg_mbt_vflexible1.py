"""BIIMG_MBT_VFLEXIBLE1 -- MI355X-native drop-in for the reference's two-stream (vital signs + CXR) MBT model with LEARNED
modality weights (builder/models/8_missing_models/biimg_mbt_vflexible1.py:17-262).

BI_VSLTIMG_MBT_V1's embeddings, frozen Swin-T features (``torch.no_grad()``, :208-209) and BimodalTransformerEncoder_MBT --
built here WITHOUT the sinusoid rows on the image stream (``use_pe=[vslt_pe, False]``, :137) and with the encoder's default
``txt_idx`` -- and the head of the "flexible" family: LayerNorm over the two CLS rows, the demographic embedding appended to each,
``fc_list`` = Linear -> LayerNorm -> ReLU -> Linear on both (:240-245), the two logits weighted by softmax(flexibleavg) over the
PRESENT modalities (an absent image is filled with -1e9 first, :247-251), summed, gathered per sample by ``missing`` (0: both,
1: vital signs alone, :253-255).  ``rmse_layer`` is always built (:157).  One image per sample (``--multiimages 0``).
"""
import torch
import torch.nn as nn

from medical_tri_modal_pilot_amd.builder.models.src.transformer.mbt_encoder import BimodalTransformerEncoder_MBT

from .bi_vsltimg_mbt_v1 import BI_VSLTIMG_MBT_V1, flat_layout

_ABSENT = torch.tensor([[False, False], [False, True]])          # rows = missing (0: both, 1: vslt only), columns = (vslt, image)


class BIIMG_MBT_VFLEXIBLE1(BI_VSLTIMG_MBT_V1):
    TRAINS_ENCODER_IN_REFERENCE = False      # frozen there too (biimg_mbt_vflexible1.py:210-211)

    def __init__(self, args):
        super().__init__(args)
        if int(getattr(args, "multiimages", 0)) == 1:
            raise NotImplementedError("BIIMG_MBT_VFLEXIBLE1 on the MI355X path: one image per sample (--multiimages 0)")
        self.output_dim = 1                                       # (:26)
        self.flexibleavg = nn.Parameter(torch.zeros(2, 1))        # (:142) root-level: first in parameters() here and there
        # same key, same position in the module order; other positional-encoding switches and the default txt_idx (:126-139)
        self.fusion_transformer = BimodalTransformerEncoder_MBT(
            batch_size=args.batch_size, n_modality=2, bottlenecks_n=4, fusion_startidx=args.mbt_fusion_startIdx,
            d_input=self.model_dim, n_layers=self.num_layers, n_head=self.num_heads, d_model=self.model_dim,
            d_ff=self.model_dim * 4, dropout=self.dropout, pe_maxlen=2500, use_pe=[False, False], mask=[True, False],
            compute_dtype=self.compute_dtype)
        classifier_dim = self.model_dim * 2
        for name in ("layer_norms_after_concat", "fc_list", "rmse_layer", "relu"):      # the reference's order (:157-163)
            if name in self._modules:
                del self._modules[name]
        self.rmse_layer = nn.Linear(classifier_dim, 1, bias=True)
        self.layer_norms_after_concat = nn.LayerNorm(self.model_dim)
        self.fc_list = nn.Sequential(nn.Linear(classifier_dim, self.model_dim, bias=True), nn.LayerNorm(self.model_dim),
                                     self.activations["relu"], nn.Linear(self.model_dim, self.output_dim, bias=True))

    def hot_parameters(self):
        skip = ("fusion_transformer.layer_norms_after_concat.", "activations.", "rmse_layer.", "img_encoder.")
        named = [(n, p) for n, p in self.named_parameters() if not n.startswith(skip)]
        return flat_layout(named, self.fusion_transformer.layer_stacks)

    def _head(self, outputs, demo_embedding, missing):
        stack = torch.stack([outputs[0][:, 0, :], outputs[1][:, 0, :]]).float()                # [2, B, 256]
        stack = self.layer_norms_after_concat(stack)
        o = self.fc_list(torch.cat([stack, demo_embedding.unsqueeze(0).expand(2, -1, -1)], dim=2))   # [2, B, 1]
        B = o.shape[1]
        miss = missing.to(o.device).long()
        w = self.flexibleavg.float().repeat(1, B).masked_fill(_ABSENT.to(o.device)[miss].permute(1, 0), -1e9)
        o = o * torch.softmax(w, dim=0).unsqueeze(2)
        cands = torch.stack([o[0] + o[1], o[0]])
        return cands[miss, torch.arange(B, device=o.device)], None, None
