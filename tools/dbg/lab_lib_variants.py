# scratch: VARIANTS = {name: [(file, old, new), ...]} for tools/lab_lib.py (patched copies of csrc/, timing experiments only)
# round 5's experiments (GELU of swin_mlp as a packed polynomial, gemm_nt fragment loads pipelined at 3 workgroups per CU) are in
# DESIGN.md section 9; what is kept here is the build with the in-kernel clock stamps (tools/dbg/swin_mlp_clock.py, dkdv64_clock.py).
VARIANTS = {
    "clock": [("Makefile", "-Wno-unused-result -mllvm", "-Wno-unused-result -DMTMP_LAB_CLOCK -mllvm")],
}
