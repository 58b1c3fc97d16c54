#!/bin/bash
# Diagnostic: kernel-trace durations of the weight-gradient launches for the shipped library and ablated builds.
#   TN_VARIANTS="A B" bash tools/dbg/prof_tn.sh      (variants as built by tools/dbg/ablate_tn.sh; "" = shipped)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_tn; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in shipped $TN_VARIANTS; do
  if [ $v = shipped ]; then unset MTMP_LIB; else export MTMP_LIB=$R/medical_tri_modal_pilot_amd/libmtmp_ab_$v.so; fi
  timeout -k 10 120 rocprofv3 --kernel-trace -d $O/$v -o x --output-format csv -- python3 $R/tools/bench_kernels.py --only gemm_tn --rounds 3 > $O/$v.log 2>&1 || { tail -5 $O/$v.log; exit 1; }
  echo "== $v"
  python3 - $O/$v/x_kernel_trace.csv <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "gemm_tn" in n or "tn_reduce" in n:
        d[(n.split("(")[0][-40:], r.get("Grid_Size", r.get("Grid_Size_X", "")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    v.sort(); print(f"   {k[0]:42s} grid {k[1]:>8s}  n {len(v):3d}  median {v[len(v)//2]:6.1f} us  min {v[0]:6.1f}")
PY
done
