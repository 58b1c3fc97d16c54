"""TIE event windows and their ragged (packed) batches.

Reference behaviour restated here (file:line in /root/reference):
  * builder/data/dataset_new.py:1969-1976  min-max normalisation of the hourly table with the training-set range;
  * builder/data/data_utils.py:28-43       ``sequenceGenerator``: the carry-forward rows of the window;
  * builder/data/dataset_new.py:1986-2001  leading / trailing hours without any measurement (``None`` entries of
                                            ``data_in_time``) are trimmed, and with ``train-missing`` the prediction
                                            time moves back by the trailing ones;
  * builder/data/dataset_new.py:2008-2026  the event list: one carried-forward "initial" event per feature whose
                                            last measurement lies before the window, then every measurement inside
                                            the window; times relative to the prediction hour (``--realtime 1``) or
                                            to the earliest event; truncation to ``--TIE-len``; zero padding.
What is NOT restated: the label / window enumeration of ``__init__`` (:1550-1935, private MIMIC index files), the
image and text branches (:2069-2172, need MIMIC-CXR files and a BioBERT h5 file).  Windows are passed in.

The batch layout is new: the reference stacks ``[TIE_len, 3]`` zero-padded tensors (default collate) and the trainer
trims them to the batch maximum (trainer.py:41-42); here a batch is the concatenation of its event lists plus the
prefix sums of their lengths, which is what ``mtmp_tie_embed_packed_fwd`` consumes (include/mtmp.h).
"""
import glob
import os
import pickle
import random
from typing import List, NamedTuple, Optional, Sequence, Tuple

import numpy as np
import torch


def tie_window(data: np.ndarray, delta: np.ndarray, data_in_time: Sequence[Optional[np.ndarray]], selected_key: int,
               rand_length: int, feature_mins: np.ndarray, feature_maxs: np.ndarray, window_size: int = 24,
               tie_len: int = 1000, realtime: int = 1, train_missing: bool = True) -> Tuple[np.ndarray, int, int]:
    """Events of the window of ``rand_length`` hours that ends at hour ``selected_key``.

    ``data`` / ``delta`` are the patient's hourly carry-forward table and hours-since-measurement table
    ``[hours, 18]`` (raw units), ``data_in_time[h]`` the ``[n, 3]`` (time, normalised value, feature) events of hour
    ``h`` or ``None``.  Returns ``(events float32 [n, 3], n, selected_key')`` with ``n <= tie_len`` and
    ``selected_key'`` the (possibly moved) prediction hour; the reference's ``txt_time`` is ``-selected_key'``."""
    rng = np.subtract(feature_maxs, feature_mins)
    norm = np.divide(np.subtract(np.asarray(data, np.float64), feature_mins), rng)                    # :1975-1976
    # sequenceGenerator (data_utils.py:28-43): only row 0 of the window is used by the TIE branch
    first = selected_key - rand_length + 1 if selected_key >= rand_length - 1 else 0
    data_row0, delta_row0 = norm[first], np.asarray(delta, np.float64)[first]
    tdl = list(data_in_time[selected_key - rand_length + 1:selected_key + 1])                         # :1979
    early, late = 0, 0
    if tdl[0] is None or tdl[-1] is None:                                                             # :1986-2001
        present = [i for i in range(len(tdl)) if tdl[i] is not None]
        if tdl[0] is None:
            early = present[0]
        elif tdl[-1] is None:
            late = rand_length - present[-1] - 1
        rand_length -= early
        if train_missing:
            selected_key -= late
        tdl = tdl[early:] if late == 0 else tdl[early:-late]
    t0 = selected_key - rand_length + 1                                                               # :2009-2012
    init = np.stack([-delta_row0 + (t0 + 1), data_row0, np.arange(18, dtype=np.float64)], axis=1)
    init = init[init[:, 0] != t0]              # features measured in the window's first hour come with that hour
    ev = np.concatenate([init] + [np.asarray(a, np.float64).reshape(-1, 3) for a in tdl if a is not None], axis=0)
    if realtime == 1:                                                                                 # :2015-2019
        ev[:, 0] -= selected_key
    else:
        ev[:, 0] -= ev[:, 0].min()
    ev = ev.astype(np.float32)[:tie_len]                                                              # :2020-2022
    return ev, int(ev.shape[0]), int(selected_key)


class PackedTie(NamedTuple):
    """Device-side packed events as the model takes them in place of the padded ``x``."""
    events: torch.Tensor        # [E, 3] float32, E >= cu_seqlens[-1] (rows past it are never read)
    cu_seqlens: torch.Tensor    # [B + 1] int32, cu_seqlens[0] = 0
    t_pad: int                  # rows per sample of the padded stream layout the embeddings are written to


class PackedTieBatch:
    """Host-side ragged batch: ``events [sum T, 3]`` (pinned when a GPU is present) + ``cu_seqlens [B + 1]``."""

    def __init__(self, events: torch.Tensor, cu_seqlens: torch.Tensor, static: torch.Tensor, txt_time: torch.Tensor):
        self.events, self.cu_seqlens, self.static, self.txt_time = events, cu_seqlens, static, txt_time

    @property
    def batch_size(self) -> int:
        return self.cu_seqlens.numel() - 1

    @property
    def input_lengths(self) -> torch.Tensor:
        return (self.cu_seqlens[1:] - self.cu_seqlens[:-1]).long()

    def to_padded(self, tie_len: int) -> torch.Tensor:
        """The reference's batch tensor ``[B, tie_len, 3]`` (zero padded, dataset_new.py:2022)."""
        x = torch.zeros(self.batch_size, tie_len, 3)
        cu = self.cu_seqlens.tolist()
        for b in range(self.batch_size):
            x[b, :cu[b + 1] - cu[b]] = self.events[cu[b]:cu[b + 1]]
        return x

    def on_device(self, device, t_pad: int, bucket: int = 0) -> PackedTie:
        """Events through the trainer's fp16 rounding (2_train.py:164) on ``device``; ``bucket`` > 0 rounds the
        event count up to a multiple (zero rows, never read) so that hipGraph replays see few distinct shapes."""
        ev = self.events.to(device, non_blocking=True).half().float()
        if bucket > 0:
            n = -(-max(ev.shape[0], 1) // bucket) * bucket
            if n != ev.shape[0]:             # padded on the device: pinning a fresh host buffer costs milliseconds per step
                pad = torch.zeros(n, 3, dtype=ev.dtype, device=ev.device)
                pad[:ev.shape[0]] = ev
                ev = pad
        return PackedTie(ev, self.cu_seqlens.to(device, non_blocking=True), int(t_pad))


def collate_packed(samples: Sequence[Tuple[np.ndarray, np.ndarray, float]]) -> PackedTieBatch:
    """``samples``: ``(events [n, 3], static [2] = (gender, age), txt_time)`` per patient window."""
    lens = [int(s[0].shape[0]) for s in samples]
    cu = torch.zeros(len(samples) + 1, dtype=torch.int32)
    cu[1:] = torch.tensor(lens, dtype=torch.int32).cumsum(0)
    events = torch.from_numpy(np.concatenate([np.asarray(s[0], np.float32).reshape(-1, 3) for s in samples], axis=0))
    static = torch.tensor(np.stack([np.asarray(s[1], np.float32) for s in samples]))
    txt_time = torch.tensor([float(s[2]) for s in samples])
    if torch.cuda.is_available():
        events, cu = events.pin_memory(), cu.pin_memory()
    return PackedTieBatch(events, cu, static, txt_time)


class SampleTieDataset(torch.utils.data.Dataset):
    """TIE windows over a directory of the reference's per-admission pickles (``data/sample_data/{train,test}``:
    keys ``data``, ``delta``, ``data_in_time``, ``age``, ``gender``).  An item is one random window of a patient,
    drawn like the reference draws it (``random.choice`` of the end hour, then of the length, :1971-1974) from the
    windows given in ``windows`` (default: every end hour with a measurement, lengths 1..window_size)."""

    def __init__(self, path: str, feature_mins, feature_maxs, window_size: int = 24, tie_len: int = 1000,
                 realtime: int = 1, train_missing: bool = True, windows: Optional[List[dict]] = None):
        self.files = sorted(glob.glob(os.path.join(path, "*.pkl")))
        if not self.files:
            raise FileNotFoundError(f"no *.pkl under {path}")
        self.fmin, self.fmax = np.asarray(feature_mins, np.float64), np.asarray(feature_maxs, np.float64)
        self.window_size, self.tie_len, self.realtime, self.train_missing = window_size, tie_len, realtime, train_missing
        self.windows = windows

    def __len__(self):
        return len(self.files)

    def _windows_of(self, index: int, p: dict) -> dict:
        if self.windows is not None:
            return self.windows[index]
        dit = p["data_in_time"]
        return {k: list(range(1, min(k + 1, self.window_size) + 1)) for k in range(len(dit)) if dit[k] is not None}

    def __getitem__(self, index: int):
        with open(self.files[index], "rb") as fh:
            p = pickle.load(fh)
        win = self._windows_of(index, p)
        key = random.choice(sorted(win))
        length = random.choice(win[key])
        ev, n, key2 = tie_window(p["data"], p["delta"], p["data_in_time"], key, length, self.fmin, self.fmax,
                                 self.window_size, self.tie_len, self.realtime, self.train_missing)
        static = np.array([1.0 if p["gender"] == "M" else 0.0, p["age"]], np.float32)                 # :1958-1962
        return ev, static, float(-key2 if self.realtime == 1 else 0)
