"""Generate tests/golden/tie_windows.npz by calling the REAL reference
``Multiple_Outbreaks_Training_Dataset.__getitem__`` (builder/data/dataset_new.py:1946-2181) on the
reference's own data/sample_data pickles.  BUILD CONTAINER ONLY.

    python tests/golden/gen/make_golden_data.py

The dataset object is made with ``object.__new__`` (its ``__init__`` walks private MIMIC index files and
opens a BioBERT h5 file); only the attributes ``__getitem__`` reads are set, and each ``_data_list`` entry
pins one (pickle, selectedKey, randLength) window, so ``random.choice`` has one element to choose from.
With ``--input-types vslt`` the image / text branches return their "missing" zeros without touching any
image file.  Import plumbing only: ``pickle5`` -> ``pickle``, empty ``h5py`` and ``torchvision.transforms``.

Stored: the sample pickles' fields the TIE construction reads (data, delta, data_in_time, age, gender,
feature_mins / feature_maxs) -- so the test needs no /root/reference -- and per case the inputs
(file index, selectedKey, randLength, TIE_len, realtime) and the reference outputs (final_seqs, static,
inputLength, txt_time, missing).
"""
import glob
import os
import pickle
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, GOLD)
import ref_shims  # noqa: E402

ref_shims.install()
sys.modules["pickle5"] = pickle
sys.modules["h5py"] = types.ModuleType("h5py")
tvt = sys.modules["torchvision.transforms"]
tvt.functional = types.ModuleType("torchvision.transforms.functional")
sys.modules["torchvision.transforms.functional"] = tvt.functional
sys.modules["torchvision"].transforms = tvt

REF = ref_shims.REF_ROOT


def main():
    sys.argv = ["2_train.py", "--input-types", "vslt", "--model", "tri_mbt_vsltcls", "--modality-inclusion",
                "train-missing_test-missing", "--output-type", "intubation", "--batch-size", "4", "--vslt-type", "TIE",
                "--model-types", "detection", "--multiimages", "0", "--berttype", "biobert", "--txt-tokenization", "bert"]
    os.chdir(REF)                      # control/config.py and builder/utils read relative paths
    from control.config import args
    from builder.data import dataset_new as D
    files = sorted(glob.glob(os.path.join(REF, "data/sample_data/train/*.pkl")))
    store = {"files": np.array([os.path.basename(f) for f in files])}
    pk = []
    for i, f in enumerate(files):
        with open(f, "rb") as fh:
            p = pickle.load(fh)
        pk.append(p)
        store[f"p{i}.data"] = np.asarray(p["data"], np.float64)
        store[f"p{i}.delta"] = np.asarray(p["delta"], np.float64)
        dit = p["data_in_time"]
        store[f"p{i}.dit_len"] = np.array([-1 if a is None else len(a) for a in dit], np.int64)
        cat = [np.asarray(a, np.float64).reshape(-1, 3) for a in dit if a is not None]
        store[f"p{i}.dit_cat"] = np.concatenate(cat) if cat else np.zeros((0, 3))
        store[f"p{i}.age"] = np.float64(p["age"])
        store[f"p{i}.male"] = np.int64(p["gender"] == "M")
    # normalisation range: the reference takes it from its training-set statistics (args.feature_mins / maxs,
    # dataset_new.py:1939-1940); here: min / max over the sample pickles (max > min enforced)
    allv = np.concatenate([np.asarray(p["data"], np.float64) for p in pk])
    fmin = allv.min(0)
    fmax = np.maximum(allv.max(0), fmin + 1.0)
    store["feature_mins"], store["feature_maxs"] = fmin, fmax
    args.feature_mins, args.feature_maxs = fmin, fmax

    ds = object.__new__(D.Multiple_Outbreaks_Training_Dataset)
    ds.window_size = args.window_size
    ds.vslt_type = "TIE"
    ds.featureidx = np.array(list(range(18)))
    ds.image_size = [args.image_size, args.image_size]
    ds.txt_token_size, ds.token_max_length = 128, 768
    ds.model_types, ds.loss_types = args.model_types, args.loss_types
    ds.neg_multi_target = [0] * 12
    cases, rng = [], np.random.RandomState(5)
    for realtime in (1, 0):
        for tie_len in (1000, 40):
            for i, p in enumerate(pk):
                T = len(p["data_in_time"])
                keys = sorted(set([T - 1, max(0, T // 2), min(T - 1, 3)] + [int(rng.randint(0, T))]))
                for key in keys:
                    for L in sorted(set([1, min(key + 1, args.window_size), int(rng.randint(1, min(key + 1, args.window_size) + 1))])):
                        win = p["data_in_time"][key - L + 1:key + 1]
                        if all(w is None for w in win):
                            continue           # the reference indexes [0] of an empty list here (never sampled by its __init__)
                        cases.append((realtime, tie_len, i, key, L))
    outs = {k: [] for k in ("case", "seq", "static", "len", "ttime", "missing")}
    for realtime, tie_len, i, key, L in cases:
        args.realtime, args.TIE_len = realtime, tie_len
        ds.time_data_array = np.zeros([tie_len, 3])
        ds._data_list = [(files[i], [key], {key: [[0]]}, {key: [L]}, 0, [], 0)]
        ds._type_list = [7]
        seq, static, target, n, img, cxr_time, tokens, tlen, ttime, missing, f_idx, taux = ds[0]
        assert float(img.abs().sum()) == 0 and tlen == 0 and cxr_time == -1
        pad = np.zeros((1000, 3), np.float32)
        pad[:tie_len] = seq.numpy()
        outs["case"].append([realtime, tie_len, i, key, L])
        outs["seq"].append(pad[:max(int(n), 1)].copy())
        outs["static"].append(static.numpy())
        outs["len"].append(int(n))
        outs["ttime"].append(float(ttime))
        outs["missing"].append(missing.numpy())
    store["case"] = np.array(outs["case"], np.int64)
    store["len"] = np.array(outs["len"], np.int64)
    store["static"] = np.stack(outs["static"]).astype(np.float32)
    store["ttime"] = np.array(outs["ttime"], np.float64)
    store["missing"] = np.stack(outs["missing"]).astype(np.float32)
    store["seq_cat"] = np.concatenate(outs["seq"]).astype(np.float32)
    store["seq_rows"] = np.array([len(s) for s in outs["seq"]], np.int64)
    np.savez_compressed(os.path.join(GOLD, "tie_windows.npz"), **store)
    print("cases", len(cases), "events", store["seq_cat"].shape, "max len", store["len"].max())


if __name__ == "__main__":
    main()
