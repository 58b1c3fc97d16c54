"""Does a graph-holding model that stays alive break the capture / replay of a later, larger one?  (round 3: a test helper that
kept its last model referenced made test_graph_full_size_training_tracks_eager segfault inside graph.replay() in the full suite)
    python tools/dbg/two_models.py [keep|drop] [n_small]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import faulthandler; faulthandler.enable()
import torch
import test_gpu_parity as T
keep = (sys.argv[1] if len(sys.argv) > 1 else "keep") == "keep"
n_small = int(sys.argv[2]) if len(sys.argv) > 2 else 3
stash = []
orig = T._product_model
def pm(*a, **k):
    r = orig(*a, **k)
    if keep:
        stash.append(r[1])
    return r
T._product_model = pm
lens = [[96, 50, 7, 1], [96, 96, 96, 96], [3, 96, 20, 64], [96, 1, 1, 2]]
for i in range(n_small):
    l, p, gs = T._loop(1, 0.0, "bf16", 4, lens)
    print("small", i, l[-1], flush=True)
T._product_model = orig
ops = T.ops.__wrapped__() if hasattr(T.ops, "__wrapped__") else None
# the full-size loop of test_graph_full_size_training_tracks_eager, graph mode
import math
from medical_tri_modal_pilot_amd.builder.trainer import get_trainer
from medical_tri_modal_pilot_amd.builder.utils.cosine_annealing_with_warmup_v2 import CosineAnnealingWarmupRestarts
from medical_tri_modal_pilot_amd.optim import FusedAdamW
torch.manual_seed(11)
args, model = orig(6, 0, "bf16", hip_graph=1, dropout=0.0, batch_size=64)
model.train()
opt = FusedAdamW(model.hot_parameters(), lr=1e-5, weight_decay=args.weight_decay)
sched = CosineAnnealingWarmupRestarts(opt, first_cycle_steps=5000, cycle_mult=1, max_lr=8e-5, min_lr=1e-6, warmup_steps=500, gamma=1.0)
crit = torch.nn.BCEWithLogitsLoss(reduction="mean")
bt = T.filler.make_batch(1234, 64, 1000, ragged=False, missing_mode="none")
d = {k: v.to(T.DEV) for k, v in bt.items() if k != "missing"}
static = torch.stack([d["gen"], d["age"]], 1)
for it in range(20):
    _, loss = get_trainer(args=args, iteration=it + 1, x=d["x"], static=static, y=d["y"], output_lengths=None, model=model,
                          logger=T._Logger(), device=torch.device(T.DEV), scheduler=sched, optimizer=opt, criterion=crit,
                          x_txt=d["txt"], x_img=d["img"], imgtxt_time=(d["img_time"], d["txt_time"]), scaler=None,
                          missing=bt["missing"], input_lengths=bt["input_lengths"], txt_lengths=d["txt_lengths"],
                          flow_type="train", reports_tokens=None, reports_lengths=None, criterion_aux=(None, None))
    print("big", it, loss, flush=True)
print("OK")
