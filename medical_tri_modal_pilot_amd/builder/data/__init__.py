"""Ragged collate / data path of the TIE (continuous-time vital-sign / lab event) stream -- SURVEY 8 f-1.

``tie_window`` restates the TIE branch of ``Multiple_Outbreaks_Training_Dataset.__getitem__``
(/root/reference builder/data/dataset_new.py:1969-2030); ``PackedTieBatch`` replaces the zero-padded
``[B, TIE_len, 3]`` batch tensor by ``(events[sum T, 3], cu_seqlens[B + 1])`` on pinned host memory."""
from .tie_dataset import (PackedTie, PackedTieBatch, SampleTieDataset, collate_packed, tie_window)  # noqa: F401
