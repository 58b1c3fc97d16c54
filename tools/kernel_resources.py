"""VGPRs / scratch / spills / occupancy of every kernel of csrc/*.hip (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_resources.py [file.hip ...] [--match attn,ln_gemm]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "medical_tri_modal_pilot_amd", "csrc")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
match = next((a.split("=", 1)[1].split(",") for a in sys.argv[1:] if a.startswith("--match=")), None)
files = args or ["attention.hip", "gemm.hip", "elementwise.hip", "swin.hip", "stem.hip", "head.hip"]
for f in files:
    extra = ["-fno-slp-vectorize"] if f == "attention.hip" else []
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result", "-mllvm",
                        "-amdgpu-mfma-vgpr-form", *extra, "-Rpass-analysis=kernel-resource-usage", "-c", f, "-o", "/dev/null"],
                       cwd=CS, capture_output=True, text=True)
    for blk in r.stderr.split("remark: Function Name: ")[1:]:
        name = subprocess.run(["c++filt", blk.split()[0]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0][:90]
        if match and not any(m in name for m in match):
            continue
        g = lambda k: re.search(k + r": (\d+)", blk).group(1)
        scr, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
        print(f"{f:16s} {name:90s} VGPR {g('VGPRs'):>3s} AGPR {g('AGPRs'):>3s} scratch {scr:>4s} spill {g('VGPRs Spill'):>3s} occ {occ}")
