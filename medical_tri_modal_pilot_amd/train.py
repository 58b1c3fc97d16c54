"""Training entry for the hot path: the optimiser / scheduler setup and the batch loop of the reference's
``2_train.py`` (:110-124 AdamW + CosineAnnealingWarmupRestarts, :141-200 the loop around ``get_trainer``), with the
build-only flags wired in:

    --synthetic 1     batches from the SURVEY §8d recipe (medical_tri_modal_pilot_amd/synthetic.py); the reference's
                      loaders need private MIMIC data and absent dependencies (out of scope)
    --ddp 1           one process per GPU (launch with ``python -m torch.distributed.run --nproc-per-node N -m
                      medical_tri_modal_pilot_amd.train ...``): RCCL gradient all-reduce through ddp.GradReducer,
                      rank-seeded batches, initial broadcast from rank 0, ``--batch-size`` is per rank
    --fused-adamw 1   optim.FusedAdamW over the flat buffers (0: torch.optim.AdamW(model.parameters()) as in the reference)
    --hip-graph 1     hipGraph replay of zero_grad + forward + backward (needs --fused-adamw 1)

    python -m medical_tri_modal_pilot_amd.train --input-types vslt_img_txt --model tri_mbt_vsltcls \\
        --modality-inclusion train-missing_test-missing --lr-init 1e-5 --batch-size 64 --epochs 1 \\
        --transformer-num-layers 6 --vslt-type TIE --imgtxt-time 1 --mbt-only-vslt 1 --synthetic 1

What is NOT here: validation / test loops over real data, tensorboard logging, checkpoint selection (2_train.py:213-376,
SURVEY §2 rows 5, 11 -- harness).  ``--iters-per-epoch`` replaces ``len(train_loader)`` for synthetic data.
"""
import math
import os
import sys
import time

import torch
import torch.distributed as dist


class _Logger:
    """The two members get_trainer touches (builder/utils/logger.py: log_lr, evaluator.add_batch) + the running loss."""

    class _Ev:
        def add_batch(self, *_):
            pass

    def __init__(self):
        self.evaluator = self._Ev()
        self.loss = 0.0
        self.lr = None

    def log_lr(self, lr, _iteration):
        self.lr = lr


def synthetic_loader(args, n_iters: int, rank: int, epoch: int):
    """n_iters batches of the 12-tuple of 2_train.py:143 (CPU tensors, like the reference's loader output)."""
    from .synthetic import make_batch
    multi = int(args.multiimages)
    for it in range(n_iters):
        bt = make_batch(1234 + 7919 * rank + 104729 * epoch + it, args.batch_size, int(args.TIE_len), ragged=True,
                        missing_mode="mixed" if "missing" in args.modality_inclusion else "none", multiimages=multi,
                        img_size=int(args.image_size), n_images=int(getattr(args, "n_images", 3)))
        static = torch.stack([bt["gen"], bt["age"]], 1)
        yield (bt["x"], static, bt["y"], bt["input_lengths"], bt["img"], bt["img_time"], bt["txt"], bt["txt_lengths"],
               bt["txt_time"], bt["missing"], None, None)


def build_training(args, device, ddp: bool):
    """model, optimizer, criterion exactly as 2_train.py:76-83,110 builds them (plus the flat-buffer AdamW / reducer)."""
    from .builder.models import get_model
    from .optim import FusedAdamW
    args.device = device
    model = get_model(args)(args).to(device)
    if ddp:
        from .ddp import broadcast_module_state
        broadcast_module_state(model, 0)
    if int(args.fused_adamw) == 1 and hasattr(model, "hot_parameters"):
        opt = FusedAdamW(model.hot_parameters(), lr=args.lr_init, weight_decay=args.weight_decay,
                         reference_params=list(model.parameters()))
        if ddp:
            from .ddp import GradReducer
            opt.reducer = GradReducer(opt.flat)
            opt.grad_scale = 1.0 / dist.get_world_size()
    else:
        if ddp:
            raise SystemExit("--ddp 1 needs --fused-adamw 1 (the reducer works on the flat gradient buffer)")
        opt = torch.optim.AdamW(model.parameters(), lr=args.lr_init, weight_decay=args.weight_decay)
    return model, opt, torch.nn.BCEWithLogitsLoss(reduction="mean")


def main(argv=None):
    from .control.config import build_parser
    from .builder.trainer import get_trainer
    from .builder.utils.cosine_annealing_with_warmup_v2 import CosineAnnealingWarmupRestarts
    parser = build_parser()
    parser.add_argument("--iters-per-epoch", type=int, default=100, help="len(train_loader) for synthetic data")
    args = parser.parse_args(argv)
    args.dir_root = os.getcwd()
    if int(args.synthetic) != 1:
        raise SystemExit("only --synthetic 1 is runnable here: the reference's data loaders need private MIMIC data "
                         "(SURVEY §2 row 9); builder/data/tie_dataset.py covers the vital-sign window construction")
    ddp = int(args.ddp) == 1
    rank, local, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    if not torch.cuda.is_available():
        raise SystemExit("training runs on an MI355X only (no CPU fallback)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    torch.manual_seed(int(args.seed_list[0]) if getattr(args, "seed_list", None) else 0)
    model, optimizer, criterion = build_training(args, device, ddp)
    n_it = int(args.iters_per_epoch)
    scheduler = CosineAnnealingWarmupRestarts(optimizer, first_cycle_steps=args.t_0 * n_it, cycle_mult=args.t_mult,
                                              max_lr=args.lr_init * math.sqrt(args.batch_size), min_lr=1e-6,
                                              warmup_steps=args.t_up * n_it, gamma=args.gamma)      # 2_train.py:118-124
    logger = _Logger()
    model.train()                                                                                    # 2_train.py:128
    iteration = 0
    for epoch in range(1, int(args.epochs) + 1):
        logger.loss, t0 = 0.0, time.perf_counter()
        for it, batch in enumerate(synthetic_loader(args, n_it, rank, epoch), 1):
            x, static, y, in_len, img, img_time, txt, txt_len, txt_time, missing, _f, _y2 = batch
            iteration += 1
            model, iter_loss = get_trainer(args=args, iteration=iteration, x=x, static=static, input_lengths=in_len, y=y,
                                           output_lengths=None, model=model, logger=logger, device=device,
                                           scheduler=scheduler, optimizer=optimizer, criterion=criterion, x_txt=txt,
                                           x_img=img, txt_lengths=txt_len, imgtxt_time=(img_time, txt_time), scaler=None,
                                           missing=missing, flow_type="train", reports_tokens=None, reports_lengths=None,
                                           criterion_aux=(None, None))
            if not math.isfinite(iter_loss):
                raise SystemExit(f"non-finite loss at iteration {iteration}")
            logger.loss += iter_loss
            if rank == 0 and it % int(args.log_iter) == 0:
                print(f"epoch {epoch} iter {it}/{n_it} loss {logger.loss / it:.5f} lr {logger.lr:.3e}", flush=True)
        if rank == 0:
            dt = time.perf_counter() - t0
            print(f"epoch {epoch}: mean loss {logger.loss / n_it:.5f}, {world * args.batch_size * n_it / dt:.1f} samples/s "
                  f"(host-resident synthetic batches, H2D inside the step)", flush=True)
    if ddp:
        dist.destroy_process_group()
    return logger.loss / max(1, n_it)


if __name__ == "__main__":
    main(sys.argv[1:])
