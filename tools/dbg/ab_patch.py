"""A/B of a module-level switch inside ONE gpurun call (boxes differ by a few %):
    python tools/dbg/ab_patch.py medical_tri_modal_pilot_amd.builder.models.src.swin_transformer._SPLIT_TAIL=False -- --steps 40 --warmup 10 --no-cpu-baseline --probe-launches 0
sets the attribute, then runs bench.py's main() with the arguments after `--` and prints ms_per_step."""
import importlib, io, json, os, sys, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sep = sys.argv.index("--") if "--" in sys.argv else len(sys.argv)
for spec in sys.argv[1:sep]:
    path, val = spec.split("=", 1)
    parts = path.split(".")
    for k in range(len(parts) - 1, 0, -1):          # longest importable module prefix, then attribute chain (classes)
        try:
            obj = importlib.import_module(".".join(parts[:k]))
            break
        except ModuleNotFoundError:
            continue
    for a in parts[k:-1]:
        obj = getattr(obj, a)
    setattr(obj, parts[-1], eval(val))
sys.argv = ["bench.py"] + sys.argv[sep + 1:]
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
rf = d.get("roofline") or {}
print("ms_per_step", round(d["ms_per_step"], 3), "median host", round(d.get("ms_per_step_median_host", 0.0), 3),
      "attn_fwd in-step us", round(1e3 * (rf.get("avg_launch_ms") or 0.0), 1), "frac", round(rf.get("frac") or 0.0, 3))
