"""One-off: evaluate the diagnostic / ablation conditionals of a kernel source with every MTMP_* switch undefined
(python tools/strip_diag.py file.hip) -- the product kernels carry no lab switches."""
import re
import sys


def ev(expr):
    """value of a preprocessor expression in which every defined(MTMP_X) is false"""
    e = re.sub(r"defined\s*\(\s*MTMP_\w+\s*\)", "0", expr)
    e = e.replace("&&", " and ").replace("||", " or ").replace("!", " not ")
    if re.search(r"[A-Za-z_]", e.replace("and", "").replace("or", "").replace("not", "")):
        return None
    return bool(eval(e))


def main(path):
    out, stack = [], []          # stack entries: [static?, currently_emitting, any_branch_taken, parent_emitting]
    for line in open(path).read().split("\n"):
        s = line.strip()
        emitting = all(f[1] for f in stack)
        m = re.match(r"#\s*(ifdef|ifndef)\s+(MTMP_\w+)", s)
        m_if = re.match(r"#\s*if\s+(.*?)(//.*)?$", s)
        if m and not (m.group(1) == "ifndef" and False):
            name = m.group(2)
            # default-value idiom (#ifndef X / #define X v / #endif) stays as it is
            val = m.group(1) == "ifndef"
            stack.append([True, val, val, emitting])
            continue
        if m_if and ev(m_if.group(1)) is not None:
            val = ev(m_if.group(1))
            stack.append([True, val, val, emitting])
            continue
        if re.match(r"#\s*(if|ifdef|ifndef)\b", s):
            stack.append([False, True, True, emitting])
            if emitting:
                out.append(line)
            continue
        m_elif = re.match(r"#\s*elif\s+(.*?)(//.*)?$", s)
        if m_elif and stack and stack[-1][0]:
            val = ev(m_elif.group(1))
            assert val is not None, line
            stack[-1][1] = (not stack[-1][2]) and val
            stack[-1][2] = stack[-1][2] or val
            continue
        if re.match(r"#\s*else\b", s) and stack and stack[-1][0]:
            stack[-1][1] = not stack[-1][2]
            stack[-1][2] = True
            continue
        if re.match(r"#\s*endif\b", s) and stack:
            f = stack.pop()
            if not f[0] and all(g[1] for g in stack):
                out.append(line)
            continue
        if emitting:
            out.append(line)
    assert not stack
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1])
