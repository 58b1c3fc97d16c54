"""Closed-form deterministic weight / input filler shared by the golden generator (build container,
real reference) and the tests (oracle, HIP path).  The code lives in the package
(medical_tri_modal_pilot_amd/synthetic.py: bench.py and ``--synthetic 1`` use it too); this module
re-exports it under the name the golden generator was written against."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from medical_tri_modal_pilot_amd.synthetic import (_hash_uniform, fill_state_dict, fill_tensor,  # noqa: E402,F401
                                                    make_batch)
