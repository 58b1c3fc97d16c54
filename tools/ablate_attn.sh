#!/bin/bash
# Diagnostic: build ablated variants of the attention forward and time them (results are WRONG by design).
cd "$(dirname "$0")/../medical_tri_modal_pilot_amd/csrc"
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form -shared"
S="attention.hip gemm.hip elementwise.hip stem.hip swin.hip head.hip error.cpp"
/opt/rocm/bin/hipcc $F -DMTMP_ABLATE_FETCH -o ../libmtmp_ab_fetch.so $S
/opt/rocm/bin/hipcc $F -DMTMP_ABLATE_EXP -o ../libmtmp_ab_exp.so $S
cd ../..
for v in "" ab_fetch ab_exp; do
  if [ -z "$v" ]; then unset MTMP_LIB; else export MTMP_LIB=$PWD/medical_tri_modal_pilot_amd/libmtmp_$v.so; fi
  echo "== variant ${v:-shipped}"; python tools/bench_kernels.py --only attn 2>&1 | grep attn_fwd
done
