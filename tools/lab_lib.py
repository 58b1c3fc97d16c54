"""A/B builds of the WHOLE library from patched copies of csrc/ (timing experiments; never shipped).  Patches live in the scratch
file tools/dbg/lab_lib_variants.py: VARIANTS = {name: [(file, old, new), ...]} (every `old` must occur exactly once).

    python tools/lab_lib.py build                 # build container: tools/dbg/_lab/<name>/libmtmp_hip.so for every variant (+ "base")
    python tools/lab_lib.py bench NAME [bench.py flags]   # GPU box: bench.py on that library (sets _lib.LIB_PATH before anything loads)
    python tools/lab_lib.py run NAME SCRIPT [flags]       # GPU box: any script of the repo on that library
"""
import os
import runpy
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "medical_tri_modal_pilot_amd", "csrc")
LAB = os.path.join(ROOT, "tools", "dbg", "_lab")


def build():
    sys.path.insert(0, os.path.join(ROOT, "tools", "dbg"))
    from lab_lib_variants import VARIANTS
    variants = dict(VARIANTS)
    variants.setdefault("base", [])
    procs = []
    for name, reps in variants.items():
        d = os.path.join(LAB, name, "csrc")
        shutil.rmtree(os.path.join(LAB, name), ignore_errors=True)
        os.makedirs(d)
        for f in os.listdir(CSRC):
            if f.endswith((".hip", ".h", ".cpp")) or f == "Makefile":
                shutil.copy(os.path.join(CSRC, f), d)
        for f, old, new in reps:
            s = open(os.path.join(d, f)).read()
            assert s.count(old) == 1, f"variant {name}: pattern occurs {s.count(old)} times in {f}: {old[:60]!r}"
            open(os.path.join(d, f), "w").write(s.replace(old, new))
        procs.append((name, subprocess.Popen(["make", "-j4", "-C", d], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for name, p in procs:
        o, _ = p.communicate()
        print(name, "ok" if p.returncode == 0 else "FAILED\n" + o[-3000:])


def bench(name, argv):
    sys.path.insert(0, ROOT)
    from medical_tri_modal_pilot_amd import _lib
    _lib.LIB_PATH = os.path.join(LAB, name, "libmtmp_hip.so")
    assert os.path.exists(_lib.LIB_PATH), _lib.LIB_PATH
    sys.argv = ["bench.py"] + argv
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")


def run(name, script, argv):
    """any script of the repo on that library: python tools/lab_lib.py run NAME tools/dbg/swin_blk_bench.py"""
    sys.path.insert(0, ROOT)
    from medical_tri_modal_pilot_amd import _lib
    _lib.LIB_PATH = os.path.join(LAB, name, "libmtmp_hip.so")
    assert os.path.exists(_lib.LIB_PATH), _lib.LIB_PATH
    sys.argv = [script] + argv
    runpy.run_path(os.path.join(ROOT, script), run_name="__main__")


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "run":
        run(sys.argv[2], sys.argv[3], sys.argv[4:])
    else:
        bench(sys.argv[2], sys.argv[3:])
