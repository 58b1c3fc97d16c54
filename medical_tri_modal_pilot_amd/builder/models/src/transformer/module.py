"""Parameter holders of the fusion transformer, same names / shapes / initialisation as
builder/models/src/transformer/module.py in the reference, so state_dicts move both ways.
The arithmetic of these blocks on the hot path is NOT here: TransformerEncoderLayer
(encoder.py) drives the fused HIP kernels directly from these parameters.
"""
import math

import torch
import torch.nn as nn
import torch.nn.init as init


class PositionalEncoding(nn.Module):
    """Sinusoid table [1, max_len, d] as a persistent buffer `pe` (reference module.py:11-32)."""

    def __init__(self, d_model: int = 512, max_len: int = 5000) -> None:
        super().__init__()
        pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        freq = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
        table = torch.zeros(max_len, d_model)
        table[:, 0::2] = torch.sin(pos * freq)
        table[:, 1::2] = torch.cos(pos * freq)
        self.register_buffer("pe", table.unsqueeze(0))

    def forward(self, length: int) -> torch.Tensor:
        return self.pe[:, :length]


class Linear(nn.Module):
    """nn.Linear under the attribute `linear`, xavier-uniform weight, zero bias (module.py:113-127)."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True) -> None:
        super().__init__()
        self.linear = nn.Linear(in_features, out_features, bias=bias)
        init.xavier_uniform_(self.linear.weight)
        if bias:
            init.zeros_(self.linear.bias)


class LayerNorm(nn.Module):
    """gamma/beta of the reference's custom LayerNorm (module.py:130-144): y = gamma*(z-mean)/(std_unbiased+eps)+beta.
    Evaluated inside mtmp_ln_gemm / mtmp_ln_bwd."""

    def __init__(self, dim: int, eps: float = 1e-6) -> None:
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim))
        self.beta = nn.Parameter(torch.zeros(dim))
        self.eps = eps


class FeedForwardUseConv(nn.Module):
    """Two k=1 Conv1d weight holders (module.py:57-80); drop1/drop2 exist for attribute parity, their
    masks are generated inside the GEMM epilogues."""

    def __init__(self, d_in: int, d_hid: int, dropout: float = 0.1) -> None:
        super().__init__()
        self.w_1 = nn.Conv1d(d_in, d_hid, 1)
        self.w_2 = nn.Conv1d(d_hid, d_in, 1)
        self.drop1 = nn.Dropout(dropout)
        self.drop2 = nn.Dropout(dropout)
