"""Instruction mix of every loop (backward branch) of one kernel in a device assembly file.

    hipcc ... --offload-device-only -S csrc/attention.hip -o /tmp/attention.s
    python tools/asm_loops.py /tmp/attention.s attn_fwd_kernelIDF16b

Per loop (label .. s_cbranch back to it): MFMA / v_exp / other VALU / ds_read / ds_write / global / s_waitcnt / s_barrier counts."""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(key) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end + 1]
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    INSTR = re.compile(r"^\s+[a-z]")
    print(f"{key}: {sum(1 for l in body if INSTR.match(l))} instructions")
    for i, l in enumerate(body):
        m = re.match(r"^\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if not m or m.group(1) not in labels or labels[m.group(1)] > i:
            continue
        seg = [x.strip().split()[0] for x in body[labels[m.group(1)]:i + 1] if INSTR.match(x)]
        c = lambda pat: sum(1 for x in seg if re.match(pat, x))
        mf, ex = c(r"v_mfma"), c(r"v_exp")
        valu = c(r"v_") - mf - ex
        print(f"  loop {m.group(1)} lines {labels[m.group(1)]}..{i}: {len(seg)} instr | mfma {mf} exp {ex} valu {valu} "
              f"(cvt_pk {c(r'v_cvt_pk')}, add {c(r'v_add_f32')}, cndmask {c(r'v_cndmask')}, mov {c(r'v_mov')}, nop {c(r's_nop')}) | "
              f"ds_read {c(r'ds_read')} ds_write {c(r'ds_write')} global {c(r'global_|buffer_')} waitcnt {c(r's_waitcnt')} barrier {c(r's_barrier')} scalar {c(r's_') }")


if __name__ == "__main__":
    main()
