"""TRI_MBT_VFLEXIBLE2 -- TRI_MBT_VFLEXIBLE with softmax temperature 10.0
(builder/models/8_missing_models/tri_mbt_vflexible2.py:279)."""
from .tri_mbt_vflexible import TRI_MBT_VFLEXIBLE


class TRI_MBT_VFLEXIBLE2(TRI_MBT_VFLEXIBLE):
    flex_temperature = 10.0
