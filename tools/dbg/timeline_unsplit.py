import sys, os
sys.path.insert(0, os.getcwd())
import medical_tri_modal_pilot_amd.builder.models.src.swin_transformer as sw
sw._SPLIT_TAIL = False
import runpy
sys.argv = ["timeline.py"]
runpy.run_path("tools/dbg/timeline.py", run_name="__main__")
