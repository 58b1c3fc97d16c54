#!/bin/bash
# Diagnostic: ablated builds of the bf16 weight-gradient product (results are WRONG by design, only the timing is read).
#   build (CPU box):  bash tools/dbg/ablate_tn.sh build       run (GPU box):  bash tools/dbg/ablate_tn.sh run
cd "$(dirname "$0")/../.."
ROOT=$PWD
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form -shared"
S="attention.hip gemm.hip elementwise.hip stem.hip swin.hip head.hip error.cpp"
V="${TN_VARIANTS:-TND_NOMMA TND_NOREAD TND_NODMA TND_NOCSUM}"   # (TN_NOMMA TN_NOCOMMIT TN_NOFETCH with -DMTMP_TN_OLD: the register-staged kernel)
if [ "$1" = build ]; then
  cd medical_tri_modal_pilot_amd/csrc
  for v in $V; do /opt/rocm/bin/hipcc $F $TN_EXTRA $(for m in ${v//+/ }; do echo -n " -DMTMP_$m"; done) -o ../libmtmp_ab_$v.so $S & done; wait
  exit 0
fi
for v in "" $V; do
  if [ -z "$v" ]; then unset MTMP_LIB; else export MTMP_LIB=$ROOT/medical_tri_modal_pilot_amd/libmtmp_ab_$v.so; fi
  echo "== variant ${v:-shipped}"; python tools/bench_kernels.py --only gemm_tn 2>&1 | grep "^gemm_tn" | grep -v blas
done
