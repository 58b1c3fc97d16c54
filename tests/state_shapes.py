"""Key -> shape of the reference TRI_MBT_VSLTCLS state_dict, from the data fixture
tests/golden/state_shapes_L2.json (dumped from the real reference module) and
expanded to any number of fusion layers."""
import json
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))


def reference_state_meta(n_layers: int):
    with open(os.path.join(_HERE, "golden", "state_shapes_L2.json")) as f:
        base = json.load(f)
    out = {}
    for k, (shape, dtype) in base.items():
        m = re.match(r"(fusion_transformer\.layer_stacks\.)(\d+)(\..*)", k)
        if m:
            if m.group(2) == "0":
                for l in range(n_layers):
                    out[f"{m.group(1)}{l}{m.group(3)}"] = (tuple(shape), dtype)
        else:
            out[k] = (tuple(shape), dtype)
    return out


def reference_state_shapes(n_layers: int, floating_only: bool = True):
    return {k: s for k, (s, d) in reference_state_meta(n_layers).items() if (not floating_only) or d.startswith("float")}
