"""Event trace of one kernel's device assembly: MFMA run lengths (Mn), spills (SW / SL), barriers, global loads (g) / stores (st) /
atomics, vmcnt waits -- to see what sits between the matrix phases.   python tools/asm_events.py file.s KERNEL_KEY"""
import re
import sys
L = open(sys.argv[1]).read().split("\n")
st = next(i for i, l in enumerate(L) if re.match(r"^_Z\w*" + re.escape(sys.argv[2]) + r"\w*:", l))
en = next(i for i in range(st, len(L)) if ".amdhsa_kernel" in L[i])
out, run = [], 0
for l in L[st:en]:
    y = l.split(";")[0].strip()
    if not y:
        continue
    op = y.split()[0]
    if op.startswith("v_mfma"):
        run += 1
        continue
    ev = None
    if op.startswith("scratch_"): ev = "S" + ("L" if "load" in op else "W")
    elif op.startswith("s_barrier"): ev = "|BAR|"
    elif op.startswith("global_load"): ev = "g"
    elif op.startswith("global_store"): ev = "st"
    elif op.startswith("global_atomic"): ev = "ATOM"
    elif op.startswith("s_waitcnt") and "vmcnt" in y: ev = "[" + y.split("vmcnt")[1].split(")")[0] + ")]"
    elif y.startswith(".LBB") and y.endswith(":"): ev = "\n" + y
    if ev:
        if run:
            out.append(f"M{run}")
            run = 0
        out.append(ev)
txt = " ".join(out)
txt = re.sub(r"(\n\.LBB\d+_\d+: )+(?=\n)", "", txt)
print(txt)
