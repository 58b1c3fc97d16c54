#!/bin/bash
# Diagnostic: build ablated variants of the LN-fused GEMM and time them (results are WRONG by design).
cd "$(dirname "$0")/../medical_tri_modal_pilot_amd/csrc"
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form -shared"
S="attention.hip gemm.hip elementwise.hip stem.hip swin.hip head.hip error.cpp"
for v in NOMMA NOEPI NOSTORE; do
  /opt/rocm/bin/hipcc $F -DMTMP_LNG_$v -o ../libmtmp_ab_$v.so $S || exit 1
done
cd ../..
for v in "" NOMMA NOEPI NOSTORE; do
  if [ -z "$v" ]; then unset MTMP_LIB; else export MTMP_LIB=$PWD/medical_tri_modal_pilot_amd/libmtmp_ab_$v.so; fi
  echo "== variant ${v:-shipped}"; python tools/bench_kernels.py --only ln_gemm 2>&1 | grep "^ln_gemm" | grep -v blas
done
