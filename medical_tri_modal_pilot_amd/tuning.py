"""Scheduling choices of the hot path, in ONE place (the A/B booleans were scattered over ops.py and trainer.py).

None of these changes a result beyond the order in which independent launches are issued or which of two equivalent launch
sequences runs (every combination is covered by the parity tests: they are bit-identical or within the documented tolerances);
each default is the faster setting as measured on MI355X -- the measurement is named beside it and lives in DESIGN.md section 9
("Built, measured, not kept").  They are module attributes so that ``tools/dbg/ab_patch.py`` can flip one inside a single gpurun
call (boxes differ by up to 5 %: an A/B across two calls says nothing); product code reads them at call time, nothing else
writes them.
"""

# ops.layer_backward: drop2's backward rides on the operand load of the dH launch (one launch and one read of d_out less).
FOLD_DROPOUT_BWD = True
# ops.layer_backward: ONE mtmp_reduce_batch per layer and stream instead of seven reduction launches of 5-13 us.
DEFER_REDUCTIONS = True
# ops.FusionStackFn.backward: issue a layer's reduction launch one layer LATE (behind the next bottleneck exchange).  Worth
# 0.05 ms while every layer ran three dense streams; with the CLS-only last layer the in-place order is faster (8.00 vs 8.06).
LATE_REDUCTIONS = False
# One launch per kernel over the streams of a fusion layer (csrc/common.hip.h, Grouped): three launches on three HIP streams held
# whole-CU workgroup slots beside the long stream's kernels (round 2: +1.2 ms / step).  bf16 build only.
GROUPED_LAUNCHES = True
# How a layer's streams are cut into launches: "small" = the vital-sign stream alone on the caller's stream, image + text together
# on a side stream (8.57-8.93 ms); "all" = one launch over all three (9.2-9.36); "none" = one launch group per stream (8.78).
GROUP_MODE = "small"
# In the layer in front of a CLS-only last layer the image / text streams' FFN runs on the four rows the exchange reads (-0.05 ms).
FFN_ROWS_BEFORE_LAST = True
# Order in which a layer's streams are issued (capture order = hardware-queue order of a replayed graph): vital signs first
# 9.23-9.37 ms, side streams first 9.51-9.65.
STREAM_ISSUE_ORDER_FWD = (0, 1, 2)
STREAM_ISSUE_ORDER_BWD = (0, 1, 2)
# builder/trainer: model.prefork() forks the frozen image encoder's stream at the head of the step, before zero_grad (-0.15 ms).
PREFORK_IMAGE_ENCODER = True
# mbt_encoder: the autograd node of the vital-sign stream's input kernel is created after the two short streams' ("long_last"), so
# that the backward issues it FIRST (autograd runs ready nodes latest-created first): the tail of the step is that stream's chain.
# "stream_order" = the order of round 4 (vital signs created first, issued last).  Measured in DESIGN section 7a.
INPUT_NODE_ORDER = "long_last"
# ops.StreamInputFn / ops.TieTimeEmbed backward: partial slabs + ONE mtmp_reduce_scatter launch that writes the parameters' slices of
# the flat gradient (the event and time embeddings' backward as one launch, the bottleneck tokens' gradient through ops.BottSink)
# instead of two reduction levels, a multi-tensor copy and the accumulation launches of the shared parameters per node.
FUSED_INPUT_TAIL = True
