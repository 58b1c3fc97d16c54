#!/bin/bash
# Diagnostic: build ablated variants of the weight-gradient GEMM and time them (results are WRONG by design).
cd "$(dirname "$0")/../medical_tri_modal_pilot_amd/csrc"
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form -shared"
S="attention.hip gemm.hip elementwise.hip stem.hip swin.hip head.hip error.cpp"
if [ "$1" = "build" ]; then
  for v in NOFETCH NOMMA NOCOMMIT; do /opt/rocm/bin/hipcc $F -DMTMP_TN_$v -o ../libmtmp_ab_tn_$v.so $S & done; wait; exit 0
fi
cd ../..
for v in "" NOFETCH NOMMA NOCOMMIT; do
  if [ -z "$v" ]; then unset MTMP_LIB; else export MTMP_LIB=$PWD/medical_tri_modal_pilot_amd/libmtmp_ab_tn_$v.so; fi
  echo "== variant ${v:-shipped}"; python tools/bench_kernels.py --only gemm_tn 2>&1 | grep "gemm_tn" | grep -v blas
done
