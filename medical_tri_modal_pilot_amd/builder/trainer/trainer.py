"""One optimisation / evaluation step with the contract of the reference's
``missing_trainer`` (builder/trainer/trainer.py:20-241): same arguments, same order of
effects (zero_grad, forward, BCE on output.squeeze(), backward, optimizer.step,
scheduler.step(iteration), logger.log_lr), same return ``(model, float loss)``.

Differences, all required to run off-CUDA-autocast and on any device:
  * no hard ``.cuda()`` (reference :77,82,84) -- tensors go to ``device``;
  * no ``torch.cuda.amp.autocast()``: the model owns its compute dtype (bf16 MFMA or fp32);
    x / img_time / txt_time still pass through fp16 rounding like 2_train.py:164 and :26-27;
  * gradient all-reduce (DDP) is waited for inside ``optimizer.step`` when a reducer is attached;
  * ``args.hip_graph == 1`` (default; needs the flat-buffer FusedAdamW, else eager): zero_grad + forward + loss +
    backward are replayed from a captured hipGraph (graph.GraphedTrainStep) -- same kernels, same order, same
    results, one launch.  The ragged trim (:41-42) then rounds max_len up to a multiple of 128 so that a
    training run needs at most a handful of graphs; the extra rows lie behind kv_len like every other pad row.
"""
import torch

from medical_tri_modal_pilot_amd import ops, tuning
from medical_tri_modal_pilot_amd.builder.data.tie_dataset import PackedTie, PackedTieBatch

GRAPH_LEN_BUCKET = 128
GRAPH_EVENT_BUCKET = 4096       # packed batches: the event count is rounded up to this for hipGraph replays


def graph_len_bucket(max_len: int, limit: int) -> int:
    """Row count a graph-replayed step pads a ragged batch to: multiples of 128 up to 1024 rows, of 256 above (graphs cannot be
    released on this ROCm -- graph.MAX_ALIVE_GRAPHS -- so the buckets a loader can visit are kept few: TIE-len 1000 -> 8,
    TIE-len 2000 -> 12; graph.GraphedTrainStep runs whatever exceeds its capture budget eagerly), never more than ``limit``
    (the batch's own padded length / --TIE-len).  The extra rows lie behind kv_len like every other pad row."""
    step = GRAPH_LEN_BUCKET if max_len <= 1024 else 2 * GRAPH_LEN_BUCKET
    return min(limit, -(-max_len // step) * step)

_TEMPLATE = {
    3: [[0., 0., 0.], [0., 0., 1.], [0., 1., 0.], [0., 1., 1.]],
    2: [[0., 0.], [0., 1.]],
}


def missing_to_num(missing: torch.Tensor, fullmodal_definition: str = "txt1_img1"):
    """rows of 0/1 flags (vslt, img, txt) -> pattern id, computed like the reference (:53-77): the batch is
    appended to the sorted template rows and ``torch.unique(dim=0, return_inverse=True)`` gives each row
    the index of its pattern.  Returns (missing_num[B] int64, missing rows used)."""
    if fullmodal_definition == "txt1":
        missing = torch.stack([missing[:, 0], missing[:, 2]]).permute(1, 0)
    elif fullmodal_definition == "img1":
        missing = missing[:, :2]
    missing = missing.detach().clone().float().cpu()
    # Fast path (same result): when every row IS one of the template rows -- first flag 0, the others 0/1 -- the sorted
    # unique rows are exactly the template, so a row's index is its binary value.  Any other row (it would become a new
    # unique row and shift the indices, as in the reference) takes the unique-based path below.
    if missing.shape[1] in (2, 3) and missing.shape[0] > 0:
        rest = missing[:, 1:]
        if bool(((missing[:, :1] == 0) & ((rest == 0) | (rest == 1))).all()):
            idx = rest[:, 0] if rest.shape[1] == 1 else 2 * rest[:, 0] + rest[:, 1]
            return idx.type(torch.LongTensor), missing
    tmpl = torch.tensor(_TEMPLATE[missing.shape[1]])
    _, inverse = torch.unique(torch.cat([tmpl, missing], dim=0), dim=0, sorted=True, return_inverse=True)
    return inverse[tmpl.shape[0]:].type(torch.LongTensor), missing


class _MissingMemo:
    """Pattern ids of the LAST device-resident ``missing`` tensor handed to the trainer.

    A device-resident ``missing`` costs a D2H copy + host sync + H2D copy per step (the pattern ids are host logic, as in
    the reference).  The memo answers only for the very same tensor OBJECT with an unchanged version counter, and it keeps
    a reference to that tensor: its storage therefore cannot go back to the caching allocator and come back as another
    batch's flags at the same address (round-2 finding: a key of (data_ptr, _version, shape, ...) alone matched a fresh
    ``missing.to(device)`` of the next batch and returned the previous batch's ids)."""

    def __init__(self):
        self.tensor = self.version = self.ctx = self.value = None

    def lookup(self, missing, ctx):
        if self.tensor is missing and self.version == missing._version and self.ctx == ctx:
            return self.value
        return None

    def store(self, missing, ctx, value):
        self.tensor, self.version, self.ctx, self.value = missing, missing._version, ctx, value

    def clear(self):
        self.__init__()


_MISSING_MEMO = _MissingMemo()


def _compute_missing_ids(args, missing, device):
    """missing flags -> pattern ids on `device` (reference trainer.py:53-77, 99-104), bounds-checked."""
    missing_num, _ = missing_to_num(missing, args.fullmodal_definition)
    if args.input_types == "vslt_txt":                                        # trainer.py:99-104
        missing_num[missing_num == 2] = 0
        missing_num[missing_num == 3] = 1
    elif args.input_types == "vslt_img":
        missing_num[missing_num == 1] = 0
        missing_num[missing_num == 3] = 1
    if missing_num.numel() and (int(missing_num.max()) > 3 or int(missing_num.min()) < 0):
        # the reference gathers all_bottleneck_stack[missing, idx_order] from FOUR candidates (mbt_encoder.py:768-776)
        raise IndexError(f"modality pattern id {int(missing_num.max())} is out of bounds for the 4 bottleneck candidates")
    return missing_num.to(device, non_blocking=True)


def _missing_ids(args, missing, device, memo=True):
    """Pattern ids of ``missing`` on ``device``; device-resident flags go through _MISSING_MEMO (same object, same version)."""
    if not (memo and missing.is_cuda):
        return _compute_missing_ids(args, missing, device)
    ctx = (str(device), args.fullmodal_definition, args.input_types)
    hit = _MISSING_MEMO.lookup(missing, ctx)
    if hit is None:
        hit = _compute_missing_ids(args, missing, device)
        _MISSING_MEMO.store(missing, ctx, hit)
    return hit


def _use_graph(args, flow_type, device, optimizer, scaler) -> bool:
    return (flow_type == "train" and int(getattr(args, "hip_graph", 0)) == 1 and torch.device(device).type == "cuda"
            and hasattr(optimizer, "flat") and scaler is None)


def _stage_bounds(args, enc, ddp: bool):
    """Layer counts (relative to the first fusion layer) at which the captured step is cut: [0, c1, ..., n_fusion].
    --graph-stages k > 0: k even groups of layers.  0 (default): one graph without DDP; under DDP TWO graphs cut behind the FIRST
    fusion layer -- the backward of layers L-1 .. 1 (and the head) is stage 0, whose gradient buckets (5/6 of the bytes at six
    layers) are all-reduced beside stage 1 = the backward of layer 0 and of the input chains (~1.1 ms at config 2, against ~0.46 ms
    for 40 MB on an 8-GPU xGMI ring); what stays exposed is the last stage's own small share.  Every cut costs ~0.06 ms of step
    time (measured with one rank, bench.py --force-ddp): three even stages exposed a third of the bytes AND paid two cuts."""
    want = int(getattr(args, "graph_stages", 0))
    auto = want <= 0
    if auto:
        want = 2 if ddp else 1
    if (enc is None or want <= 1 or enc.resbottle or not getattr(enc, "supports_segments", False)
            or getattr(args, "vslt_type", "TIE") == "QIE"):
        return [0, 0]
    n_fl = enc.n_layers - min(max(enc.fusion_idx, 0), enc.n_layers)
    if auto:
        return [0, 1, n_fl] if n_fl >= 2 else [0, n_fl]
    want = min(want, n_fl)
    if want <= 1:
        return [0, n_fl]
    return sorted(set([0, n_fl] + [round(n_fl * k / want) for k in range(1, want)]))


def _staged_step(model, enc, optimizer, criterion, run_model, bounds):
    """The stage callables of graph.GraphedTrainStep for a step cut at ``bounds``: stage 0 = zero_grad, forward, loss
    and the backward of the head and the LAST group of fusion layers, down to the stream buffers in front of that
    group (mbt_encoder.py segment_boundaries); stage k = the backward of the next group down; the last stage also
    runs the input-side backward (stream inputs, projections, TIE embedding)."""
    n_pre = min(max(enc.fusion_idx, 0), enc.n_layers)
    n_stage = len(bounds) - 1

    flat = optimizer.flat

    def partial(outputs, grad_outputs, bnd, prm):
        """Backward from ``outputs`` that STOPS at the stream buffers ``bnd`` (returns their gradients) and accumulates
        into the parameters ``prm`` -- torch.autograd.grad, not backward(inputs=...): the latter executes the
        producer node of a non-leaf input (it retains its grad through a hook on that node), i.e. the next stage's
        work.  Gradients the kernels wrote straight into the flat buffer come back as None."""
        res = torch.autograd.grad(outputs, list(bnd) + prm, grad_outputs, allow_unused=True, retain_graph=True)
        for q, g in zip(prm, res[len(bnd):]):
            if g is not None:
                flat.add_grad(q, g)
        return res[:len(bnd)]

    def stage0(t, carry):
        if tuning.PREFORK_IMAGE_ENCODER and hasattr(model, "prefork"):
            model.prefork(t["x_img"])
        optimizer.zero_grad()
        step_loss = ops.bce_with_logits(criterion, run_model(t), t["final_target"])
        carry["loss"] = step_loss.detach()
        gs_ = getattr(model, "_mtmp_graph_step", None)
        if gs_ is not None:
            gs_.publish_loss(step_loss)
        bnds = carry["bnds"] = list(enc.segment_boundaries)
        if len(bnds) != n_stage - 1:          # the encoder did not cut (torch input chain, resbottle, ...): one backward
            carry["bnds"] = []
            torch.autograd.backward([step_loss], [ops.unit_grad(step_loss)])
            return
        prm = model.backward_stage_params(n_pre + bounds[-2], n_pre + bounds[-1], head=True)
        carry["g"] = partial([step_loss], [ops.unit_grad(step_loss)], bnds[-1], prm)

    def later(k):
        def stage(t, carry):
            bnds = carry["bnds"]
            if not bnds:
                return
            j = len(bnds) - k                  # carry["g"] = gradients of bnds[j]; this stage takes them down to bnds[j-1]
            src = [(z, g) for z, g in zip(bnds[j], carry["g"]) if g is not None]
            outs, grads = [z for z, _ in src], [g for _, g in src]
            if j == 0:
                torch.autograd.backward(outs, grads)
            else:
                prm = model.backward_stage_params(n_pre + bounds[j], n_pre + bounds[j + 1], head=False)
                carry["g"] = partial(outs, grads, bnds[j - 1], prm)
        return stage

    return [stage0] + [later(k) for k in range(1, n_stage)]


def missing_trainer(args, iteration, train_x, static_x, input_lengths, train_y, model, logger, device,
                    scheduler=None, optimizer=None, criterion=None, scaler=None, flow_type=None, output_lengths=None,
                    seq_lengths=None, x_img=None, x_txt=None, txt_lengths=None, imgtxt_time=None, missing=None,
                    reports_tokens=None, reports_lengths=None, criterion_aux=None):
    img_time, txt_time = imgtxt_time
    # fp16 rounding of the event / time inputs (trainer.py:26-27, 2_train.py:164).  A device-resident fp32 tensor of a step
    # that will be replayed from a hipGraph is rounded by the copy into the graph's static buffers instead
    # (GraphedTrainStep.run round_fp16): two eager launches per tensor less in front of every step.
    deferred = set()
    defer_ok = flow_type == "train" and _use_graph(args, flow_type, device, optimizer, scaler) and output_lengths is None

    def fp16_round(t, key):
        if (defer_ok and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
                and torch.device(device).index in (None, t.device.index)):
            deferred.add(key)
            return t
        return t.half().float().to(device, non_blocking=True)
    img_time = fp16_round(img_time, "img_time")
    txt_time = fp16_round(txt_time, "txt_time")
    cu_seqlens = None
    if isinstance(train_x, PackedTieBatch):
        # Ragged batch (builder/data): the events travel packed, there is nothing to trim; the padded
        # row count of the stream buffers is the batch maximum (bucketed for graph replays like the trim below).
        graphed = _use_graph(args, flow_type, device, optimizer, scaler)
        max_len = int(torch.max(input_lengths))
        if graphed:
            max_len = graph_len_bucket(max_len, int(args.TIE_len))
        pk = train_x.on_device(device, max_len, GRAPH_EVENT_BUCKET if graphed else 0)   # fp16 rounding inside
        data, cu_seqlens, t_pad = pk.events, pk.cu_seqlens, pk.t_pad
    elif args.vslt_type == "carryforward":
        train_x = train_x.permute(1, 0, 2, 3)
        data = train_x[0].half().float().to(device, non_blocking=True)
    else:
        # ragged trim, trainer.py:41-42 (at least one row: a batch of all-empty windows keeps one pad event behind kv_len --
        # pad rows feed nothing -- instead of zero-length launches)
        max_len = max(1, int(torch.max(input_lengths)))
        if _use_graph(args, flow_type, device, optimizer, scaler):
            max_len = graph_len_bucket(max_len, train_x.shape[1])
        data = fp16_round(train_x[:, :max_len, :], "data")                    # 2_train.py:164
    if "rmse" in args.auxiliary_loss_type:
        final_target = train_y[0].float().to(device, non_blocking=True)
    else:
        final_target = train_y.float().to(device, non_blocking=True)
    missing_num = _missing_ids(args, missing, device)
    static_x = static_x.permute(1, 0)
    age = static_x[1].float().to(device, non_blocking=True)
    gender = static_x[0].float().to(device, non_blocking=True)
    x_txt = x_txt.to(device, non_blocking=True)
    x_img = x_img.to(device, non_blocking=True)
    input_lengths = input_lengths.to(device, non_blocking=True)
    txt_lengths = txt_lengths.to(device, non_blocking=True)
    feasible = None if output_lengths is None else output_lengths.type(torch.IntTensor).to(device, non_blocking=True)

    packed_extra = {}
    if cu_seqlens is not None:       # t_pad rides in the graph signature as the shape of an empty tensor
        packed_extra = dict(cu_seqlens=cu_seqlens, t_pad_marker=torch.empty(t_pad, 0, device=device))

    red = getattr(optimizer, "reducer", None)
    if red is not None and not getattr(red, "_mtmp_streams_set", False) and hasattr(model, "fusion_transformer"):
        red.extra_streams = list(model.fusion_transformer._side_streams(torch.device(device)) or [])
        red._mtmp_streams_set = True

    def run_model(t=None):
        if t is None:
            t = dict(data=data, age=age, gender=gender, input_lengths=input_lengths, x_txt=x_txt,
                     txt_lengths=txt_lengths, x_img=x_img, missing_num=missing_num, img_time=img_time,
                     txt_time=txt_time, **packed_extra)
        x_in = t["data"] if "cu_seqlens" not in t else PackedTie(t["data"], t["cu_seqlens"], t["t_pad_marker"].shape[0])
        out, _, _ = model(x_in, None, None, None, None, t["age"], t["gender"], t["input_lengths"], t["x_txt"],
                          t["txt_lengths"], t["x_img"], t["missing_num"], feasible, t["img_time"], t["txt_time"],
                          flow_type, reports_tokens, reports_lengths)
        return out.squeeze()

    if flow_type == "train" and _use_graph(args, flow_type, device, optimizer, scaler) and feasible is None:
        from medical_tri_modal_pilot_amd.graph import GraphedTrainStep
        gs = getattr(model, "_mtmp_graph_step", None)
        if gs is None or gs.device != data.device:
            gs = model._mtmp_graph_step = GraphedTrainStep(data.device, max_graphs=int(getattr(args, "hip_graph_max", 12)),
                                                           fallback=bool(int(getattr(args, "hip_graph_fallback", 0))))
        # Data-parallel steps are captured as a few graphs cut at layer boundaries: the all-reduce of the gradient
        # buckets a stage completed starts right behind its replay and overlaps the next stage (ddp.py, staged mode).
        enc = getattr(model, "fusion_transformer", None)
        bounds = _stage_bounds(args, enc, red is not None)
        if red is not None:
            red.staged = True
        if enc is not None and getattr(enc, "supports_segments", False):
            enc.graph_segments = bounds[1:-1]
        inputs = dict(data=data, age=age, gender=gender, input_lengths=input_lengths, x_txt=x_txt,
                      txt_lengths=txt_lengths, x_img=x_img, missing_num=missing_num, img_time=img_time,
                      txt_time=txt_time, final_target=final_target, **packed_extra)
        if len(bounds) <= 2:
            def fwd_bwd(t):
                if tuning.PREFORK_IMAGE_ENCODER and hasattr(model, "prefork"):
                    model.prefork(t["x_img"])            # the image encoder's stream forks at the head of the step
                optimizer.zero_grad()
                step_loss = ops.bce_with_logits(criterion, run_model(t), t["final_target"])
                gs.publish_loss(step_loss)           # the host takes the value from here (gs.wait_loss below)
                torch.autograd.backward([step_loss], [ops.unit_grad(step_loss)])
                ops.mark("bwd.e")                    # (tools/dbg/timeline.py; nothing is launched unless marks are enabled)
                return step_loss.detach()
            loss = gs.run(inputs, fwd_bwd, optimizer.flat.params, reducer=red, round_fp16=deferred)
        else:
            loss = gs.run(inputs, _staged_step(model, enc, optimizer, criterion, run_model, bounds),
                          optimizer.flat.params, reducer=red, round_fp16=deferred)
        optimizer.step()
        scheduler.step(iteration)
        logger.log_lr(scheduler.get_lr()[0], iteration)
        # The reference's loss.item() (trainer.py:128), without waiting for the backward and the optimizer step that are still
        # running: the value was copied to pinned memory right behind the forward pass.  What the caller enqueues next is
        # stream-ordered behind this step as always; only the host no longer idles the GPU between two steps.
        return model, gs.wait_loss()
    elif flow_type == "train":
        if red is not None:
            red.staged = False
        if getattr(getattr(model, "fusion_transformer", None), "supports_segments", False):
            model.fusion_transformer.graph_segments = None
        optimizer.zero_grad()
        output = run_model()
        loss = ops.bce_with_logits(criterion, output, final_target)
        torch.autograd.backward([loss], [ops.unit_grad(loss)])
        optimizer.step()
        scheduler.step(iteration)
        logger.log_lr(scheduler.get_lr()[0], iteration)
    else:
        with torch.no_grad():
            output = run_model()
            loss = ops.bce_with_logits(criterion, output, final_target)
            output = torch.sigmoid(output)
        logger.evaluator.add_batch(final_target, output)
    return model, loss.item()
