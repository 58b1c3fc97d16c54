"""One-line-per-slot view of a loop of a kernel's device assembly: python tools/asm_stream.py file.s KERNEL_KEY LOOP_LABEL
(M<dst> = MFMA, e = v_exp, a = v_add_f32, c = v_cvt_pk, v = other VALU, R / W = ds_read / ds_write, G = global load, n = s_nop)."""
import re
import sys


def main():
    path, key, label = sys.argv[1:4]
    L = open(path).read().split("\n")
    st = next(i for i, l in enumerate(L) if re.match(r"^_Z\w*" + re.escape(key) + r"\w*:", l))
    body = L[st:]
    lab = next(i for i, l in enumerate(body) if l.startswith(label + ":"))
    end = next(i for i in range(lab, len(body)) if re.match(r"^\s+s_c?branch\w*\s+" + re.escape(label) + r"\b", body[i]))
    out = []
    for x in body[lab:end + 1]:
        x = x.split(";")[0].strip()
        if not x or x.endswith(":"):
            continue
        op = x.split()[0]
        if op.startswith("v_mfma"): out.append("\n M" + x.split()[1].rstrip(","))
        elif op.startswith("v_exp"): out.append("e")
        elif op.startswith("v_add_f32"): out.append("a")
        elif op.startswith("v_cvt_pk"): out.append("c")
        elif op.startswith("ds_read"): out.append("R")
        elif op.startswith("ds_write"): out.append("W")
        elif op.startswith("global_load"): out.append("G")
        elif op.startswith("s_waitcnt"): out.append("[" + "".join(x.split()[1:]) + "]")
        elif op.startswith("s_barrier"): out.append("[BARRIER]")
        elif op.startswith("s_nop"): out.append("n")
        elif op.startswith("v_"): out.append("v")
        elif op.startswith("s_"): out.append("s")
    print(" ".join(out))


if __name__ == "__main__":
    main()
