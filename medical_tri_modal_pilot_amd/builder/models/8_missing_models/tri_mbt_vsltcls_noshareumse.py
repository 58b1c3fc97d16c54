"""TRI_MBT_VSLTCLS_NOSHAREUMSE -- MI355X-native drop-in for the reference's sibling of the tri-modal model
(builder/models/8_missing_models/tri_mbt_vsltcls_noshareumse.py:17-266).

Same fusion encoder (TrimodalTransformerEncoder_MBT), same head, same forward signature and state_dict keys as the
reference class; it differs from TRI_MBT_VSLTCLS in its embedding chains only:

  * UMSE chains WITHOUT LayerNorm and with a second projection: ``Linear(1, 256) -> ReLU -> Linear(256, 256, bias=False)``
    for the value (``ie_vslt``) and the event time (``ie_time``), reference :61-71;
  * the image / report times have their OWN chains of that shape (``ie_time_img`` / ``ie_time_txt``, :72-81, :226-227)
    instead of sharing ``ie_time`` ("no share");
  * ``ie_demo = Linear(2, 256) -> ReLU`` (no LayerNorm, :83-86);
  * the image encoder starts from random weights (``swin_t_m(weights=None)``, :111-112) -- same module here.

The 256 x 256 projections are GEMMs over all events (config 2: 64,000 x 256 x 256 twice): they run on the HIP GEMM path
(ops.LinearFn: mtmp_gemm_nt forward and dx, mtmp_gemm_tn weight gradient); the scalar -> 256 affine + ReLU in front is
elementwise torch code.  The rest of the step is the parent's.
"""
import torch
import torch.nn as nn

from medical_tri_modal_pilot_amd import ops
from medical_tri_modal_pilot_amd.builder.data.tie_dataset import PackedTie

from .tri_mbt_vsltcls import TRI_MBT_VSLTCLS


class TRI_MBT_VSLTCLS_NOSHAREUMSE(TRI_MBT_VSLTCLS):
    head_fusable = False          # ie_demo has no LayerNorm: the head's torch form (parent forward) runs

    def _make_embeddings(self, args) -> bool:
        def chain(n_in):
            return nn.Sequential(nn.Linear(n_in, self.model_dim), nn.ReLU(inplace=True),
                                 nn.Linear(self.model_dim, self.model_dim, bias=False))

        if args.vslt_type == "carryforward":
            self.vslt_enc = chain(self.num_nodes)
            vslt_pe = True
        elif args.vslt_type in ("TIE", "QIE"):
            vslt_pe = False
            self.ie_vslt = chain(1)
        else:
            raise ValueError(args.vslt_type)
        self.ie_time = chain(1)
        self.ie_time_txt = chain(1)
        self.ie_time_img = chain(1)
        self.ie_feat = nn.Embedding(20, self.model_dim)
        self.ie_demo = nn.Sequential(nn.Linear(2, self.model_dim), nn.ReLU(inplace=True))
        if args.vslt_type == "QIE":
            raise NotImplementedError("TRI_MBT_VSLTCLS_NOSHAREUMSE on the MI355X path: --vslt-type TIE or carryforward")
        return vslt_pe

    def _chain(self, seq, v, dt):
        """seq = Linear(1, 256) -> ReLU -> Linear(256, 256, bias=False) applied to scalars v [...]: [..., 256] in dt."""
        h = torch.relu(v.unsqueeze(-1) * seq[0].weight[:, 0] + seq[0].bias)            # [..., 256] fp32, elementwise
        if not h.is_cuda:
            return torch.nn.functional.linear(h, seq[2].weight).to(dt)
        return ops.LinearFn.apply(h, seq[2].weight, None, dt)

    def _joint_embeddings(self, x, img_time, txt_time, dt):
        return None                       # three time chains of their own: nothing is shared

    def _vslt_embedding(self, x, dt):
        if isinstance(x, PackedTie):
            raise NotImplementedError("TRI_MBT_VSLTCLS_NOSHAREUMSE takes the padded event tensor (no packed batches)")
        feat = x[:, :, 2].to(torch.int64)                                             # :187 x[:,:,2].type(IntTensor)
        return (self._chain(self.ie_vslt, x[:, :, 1], dt) + self._chain(self.ie_time, x[:, :, 0], dt)
                + self.ie_feat(feat).to(dt))

    def _time_embeddings(self, img_time, txt_time, demo_embedding, dt):
        it = self._chain(self.ie_time_img, img_time, dt) + self.ie_feat.weight[18].to(dt)
        tt = self._chain(self.ie_time_txt, txt_time, dt) + self.ie_feat.weight[19].to(dt)
        return it, tt
