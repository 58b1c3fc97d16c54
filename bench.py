"""Headline benchmark: tri-modal training samples/s (BASELINE.json metric) on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one full optimisation step of TRI_MBT_VSLTCLS through the reference's trainer
contract (zero_grad, forward, BCE, backward, AdamW, scheduler) on one synthetic batch that is
already resident in HBM.  Workload = BASELINE.json configs[1]: vslt_img_txt, 6 layers, d_model
256, per-GPU batch 64, TIE-len 1000 (full-length events -> N_v = 1005), one 224x224 CXR, 128
text tokens, bf16 MFMA build, dropout 0.1 (the reference default), random-init weights.
Weak scaling: every rank runs its own batch of 64; gradients are all-reduced over RCCL.

The JSON line also carries
  roofline     -- the dominant kernel (key-masked attention forward, vslt stream): algorithmic
                  FLOPs 4*B*H*N^2*64 per launch / average launch time measured with HIP events
                  on the launch stream inside the timed steps, against the 2.5 PFLOP/s dense bf16 peak;
  cpu_baseline -- the CPU oracle (fp32 PyTorch restatement of the reference, golden-pinned) timed
                  on this host's cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

B_PER_GPU, TIE_LEN, LAYERS = 64, 1000, 6
PEAK_BF16_TFLOPS = 2500.0       # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def make_args(dtype: str, dropout: float, hip_graph: int = 1):
    from medical_tri_modal_pilot_amd.control.config import parse_args
    return parse_args(["--input-types", "vslt_img_txt", "--model", "tri_mbt_vsltcls", "--modality-inclusion",
                       "train-missing_test-missing", "--lr-init", "1e-5", "--output-type", "intubation",
                       "--batch-size", str(B_PER_GPU), "--transformer-num-layers", str(LAYERS), "--vslt-type", "TIE",
                       "--model-types", "detection", "--imgtxt-time", "1", "--mbt-only-vslt", "1", "--multiimages", "0",
                       "--dropout", str(dropout), "--compute-dtype", dtype, "--hip-graph", str(hip_graph)])


def cpu_baseline(sample_b: int = 8, timed: int = 2):
    """Oracle (port of the reference's CPU path) on a bounded sample of the same workload."""
    import filler
    from oracle import tri_mbt_oracle as O
    from tests.state_shapes import reference_state_shapes
    torch.manual_seed(0)
    sd = {}
    for k, s in reference_state_shapes(LAYERS).items():
        sd[k] = filler.fill_tensor(k, torch.zeros(s))
    sd["fusion_transformer.positional_encoding.pe"] = O.sinusoid_table(2500, 256).unsqueeze(0)
    bt = filler.make_batch(1234, sample_b, TIE_LEN, ragged=False, missing_mode="none")
    tr = O.OracleTrainer(sd, O.Cfg(n_layers=LAYERS, dropout=0.1), lr_init=1e-5, batch_size=sample_b, iters_per_epoch=100)
    tr.step(bt, 1)                                   # warm-up
    ts = []
    for i in range(timed):
        t0 = time.perf_counter()
        tr.step(bt, 2 + i)
        ts.append(time.perf_counter() - t0)
    t = sorted(ts)[len(ts) // 2]
    return {"value": sample_b / t, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle full train step (fwd+BCE+bwd+AdamW), fp32, B={sample_b} of the config-2 batch "
                      f"(T={TIE_LEN}, L={LAYERS}), 1 warm-up + {timed} timed, median {t:.2f} s/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--hip-graph", type=int, default=1, choices=[0, 1])
    ap.add_argument("--workload", default="full", choices=["full", "ragged"],
                    help="full = BASELINE configs[1] (every series at TIE-len, all modalities; the headline number); "
                         "ragged = configs[3] shape: len ~ U{3..T}, mixed missing modalities (SURVEY 8d)")
    ap.add_argument("--packed", type=int, default=0, choices=[0, 1],
                    help="feed the vital-sign events as the ragged PackedTieBatch of builder/data (SURVEY 8 f-1)")
    ap.add_argument("--probe-steps", type=int, default=5, help="eager steps after the timed region that time "
                    "the roofline kernel with HIP events (only when the timed region replays a hipGraph)")
    a = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # MTMP_FORCE_DDP=1: run the RCCL reducer even with one rank (exercises the N>1 code path on a 1-GPU box)
    ddp = world > 1 or bool(os.environ.get("MTMP_FORCE_DDP"))
    if ddp and "MASTER_ADDR" not in os.environ:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    if ddp:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import filler
    from medical_tri_modal_pilot_amd import ops
    from medical_tri_modal_pilot_amd.builder.models import get_model
    from medical_tri_modal_pilot_amd.builder.trainer import get_trainer
    from medical_tri_modal_pilot_amd.builder.utils.cosine_annealing_with_warmup_v2 import CosineAnnealingWarmupRestarts
    from medical_tri_modal_pilot_amd.optim import FusedAdamW
    from medical_tri_modal_pilot_amd.ddp import GradReducer, broadcast_module_state

    args = make_args(a.dtype, a.dropout, a.hip_graph)
    args.device = dev
    torch.manual_seed(412)
    model = get_model(args)(args).to(dev)
    if ddp:
        broadcast_module_state(model, 0)
    model.train()
    if os.environ.get("MTMP_NO_OVERLAP"):            # experiment switch: all three modality streams on one HIP stream
        model.fusion_transformer.overlap_streams = False
    opt = FusedAdamW(model.hot_parameters(), lr=args.lr_init, weight_decay=args.weight_decay)
    if ddp:
        opt.reducer = GradReducer(opt.flat)
        opt.grad_scale = 1.0 / world
    sched = CosineAnnealingWarmupRestarts(opt, first_cycle_steps=args.t_0 * 100, cycle_mult=args.t_mult,
                                          max_lr=args.lr_init * math.sqrt(args.batch_size), min_lr=1e-6,
                                          warmup_steps=args.t_up * 100, gamma=args.gamma)
    crit = torch.nn.BCEWithLogitsLoss(reduction="mean")

    class Log:
        class Ev:
            def add_batch(self, *_):
                pass
        evaluator = Ev()

        def log_lr(self, *_):
            pass

    ragged = a.workload == "ragged"
    bt = filler.make_batch(1234 + rank, B_PER_GPU, TIE_LEN, ragged=ragged, missing_mode="mixed" if ragged else "none")
    d = {k: v.to(dev) for k, v in bt.items() if k != "missing"}
    static = torch.stack([d["gen"], d["age"]], 1)
    x_in = d["x"]
    if a.packed:
        from medical_tri_modal_pilot_amd.builder.data import collate_packed
        x_in = collate_packed([(bt["x"][b, :int(n)].numpy(), static[b].cpu().numpy(), float(bt["txt_time"][b]))
                               for b, n in enumerate(bt["input_lengths"])])
    kw = dict(args=args, x=x_in, static=static, y=d["y"], output_lengths=None, model=model, logger=Log(),
              device=dev, scheduler=sched, optimizer=opt, criterion=crit, x_txt=d["txt"], x_img=d["img"],
              imgtxt_time=(d["img_time"], d["txt_time"]), scaler=None, missing=bt["missing"], flow_type="train",
              reports_tokens=None, reports_lengths=None, criterion_aux=(None, None))
    in_len_host = bt["input_lengths"]            # lengths stay on the host like the reference loader's

    # HIP-event instrumentation of the dominant kernel (vslt-stream attention forward)
    events = []
    raw_attn_fwd = ops.attn_fwd
    record = {"on": False}

    def timed_attn_fwd(qkv, kv_len, res=None):
        if record["on"] and qkv.shape[1] == TIE_LEN + 5:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = raw_attn_fwd(qkv, kv_len, res)
            e1.record()
            events.append((e0, e1))
            return out
        return raw_attn_fwd(qkv, kv_len, res)

    ops.attn_fwd = timed_attn_fwd

    # the same for the two other heavy kernels of the vital-sign stream (reported under "roofline_more"):
    # attention backward (dQ + dK/dV launches of one call) and the weight-gradient GEMMs
    more = {"attn_bwd": [], "gemm_tn": []}
    raw_attn_bwd, raw_gemm_tn = ops.attn_bwd, ops.gemm_tn

    def timed_attn_bwd(qkv, *rest):
        if record["on"] and qkv.shape[1] == TIE_LEN + 5:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = raw_attn_bwd(qkv, *rest)
            e1.record()
            more["attn_bwd"].append((e0, e1, 10.0 * qkv.shape[0] * 4 * qkv.shape[1] * qkv.shape[1] * 64))
            return out
        return raw_attn_bwd(qkv, *rest)

    def timed_gemm_tn(dy2d, x2d, *rest, **kws):
        if record["on"] and dy2d.shape[0] >= B_PER_GPU * TIE_LEN:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = raw_gemm_tn(dy2d, x2d, *rest, **kws)
            e1.record()
            more["gemm_tn"].append((e0, e1, 2.0 * dy2d.shape[0] * dy2d.shape[1] * x2d.shape[1]))
            return out
        return raw_gemm_tn(dy2d, x2d, *rest, **kws)

    ops.attn_bwd, ops.gemm_tn = timed_attn_bwd, timed_gemm_tn

    # host-side enqueue time: from step start until the trainer blocks in loss.item()
    enq = {"t0": 0.0, "sum": 0.0, "n": 0}
    raw_item = torch.Tensor.item

    def timed_item(self):
        if record["on"] and enq["t0"] > 0:
            enq["sum"] += time.perf_counter() - enq["t0"]
            enq["n"] += 1
            enq["t0"] = 0.0
        return raw_item(self)

    torch.Tensor.item = timed_item

    def step(it):
        enq["t0"] = time.perf_counter()
        return get_trainer(iteration=it, input_lengths=in_len_host, txt_lengths=d["txt_lengths"], **kw)[1]

    for i in range(a.warmup):
        step(i + 1)
    record["on"] = True
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = 0.0
    for i in range(a.steps):
        loss = step(a.warmup + i + 1)
        if os.environ.get("MTMP_PRINT_LOSS"):
            print(f"step {a.warmup + i + 1} loss {loss:.6f}", file=sys.stderr)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    record["on"] = False
    host_ms = 1e3 * enq["sum"] / max(1, enq["n"])
    gs = getattr(model, "_mtmp_graph_step", None)
    graphed = gs is not None and gs.replays >= a.steps
    if graphed and rank == 0:
        # events cannot time one kernel inside a replayed graph: time the same kernel on the same workload in
        # eager steps right after the timed region (its rocprofv3 average covers both kinds of launch)
        events.clear()
        more["attn_bwd"].clear()
        more["gemm_tn"].clear()
        args.hip_graph = 0
        record["on"] = True
        for i in range(a.probe_steps):
            step(a.warmup + a.steps + i + 1)
        torch.cuda.synchronize()
        record["on"] = False
        args.hip_graph = a.hip_graph
    if world > 1:
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    if not math.isfinite(loss):
        raise SystemExit(f"non-finite loss {loss}")

    if rank == 0:
        n_tok = TIE_LEN + 5
        k_ms = sum(e0.elapsed_time(e1) for e0, e1 in events) / max(1, len(events))
        flops = 4.0 * B_PER_GPU * 4 * n_tok * n_tok * 64
        achieved = flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        # HBM traffic of the same kernel from the PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate rocprofv3
        # runs of this script -- counters cannot be read from inside it); committed under profiles/ with its method
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if a.dtype == "bf16" and os.path.exists(tpath):
            with open(tpath) as fh:
                tj = json.load(fh)
            traffic, traffic_src = tj.get("traffic_bytes_per_launch"), f"profiles/roofline_traffic.json ({tj.get('round')})"
        out = {
            "metric": "tri-modal training samples/s (fwd+bwd+AdamW step, per-GPU batch 64)",
            "value": world * B_PER_GPU * a.steps / dt, "unit": "samples/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "host_enqueue_ms_per_step": host_ms, "hip_graph": bool(graphed), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic (SURVEY 8d recipe), random-init weights",
            "config": {"workload": ("BASELINE configs[1]: vslt_img_txt tri_mbt_vsltcls, 6 layers, d_model 256, "
                                    "batch 64/GPU, TIE-len 1000 (N_v=1005), 224x224 CXR, 128-tok text, "
                                    f"dropout {a.dropout}, mbt-only-vslt 1, imgtxt-time 1")
                                   + (" -- RAGGED variant (configs[3] shape): len ~ U{3..1000}, mixed missing modalities;"
                                      " roofline flops still count the dense N^2" if ragged else "")
                                   + (" -- events fed as PackedTieBatch" if a.packed else ""),
                       "global_batch": world * B_PER_GPU, "parallelism": f"dp{world}", "final_loss": loss},
            "roofline": {"bound": "mfma", "kernel": "attn_fwd_kernel<bf16> (vslt stream, N=1005)" if a.dtype == "bf16"
                         else "attn_fwd_kernel<float>", "achieved": achieved, "peak": PEAK_BF16_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src,
                         "launches_timed": len(events), "avg_launch_ms": k_ms, "flops_per_launch": flops,
                         "timed_in": "eager probe steps after the graph-replay region" if graphed else "timed steps"},
        }
        names = {"attn_bwd": "attn_bwd_dq_kernel + attn_bwd_dkdv_kernel (vslt stream; algorithmic flops = 2.5 x forward)",
                 "gemm_tn": "gemm_tn_tr_kernel + tn_reduce_kernel (vslt-stream weight gradients)"}
        out["roofline_more"] = []
        for key, evs in more.items():
            ms = sum(e0.elapsed_time(e1) for e0, e1, _ in evs)
            fl = sum(f for _, _, f in evs)
            if evs and ms > 0:
                tf = fl / (ms * 1e-3) / 1e12
                out["roofline_more"].append({"bound": "mfma", "kernel": names[key], "achieved": tf, "peak": PEAK_BF16_TFLOPS,
                                             "unit": "TFLOP/s", "frac": tf / PEAK_BF16_TFLOPS, "launches_timed": len(evs),
                                             "avg_launch_ms": ms / len(evs), "flops_per_launch": fl / len(evs)})
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
