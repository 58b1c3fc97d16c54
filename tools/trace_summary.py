"""Summarise a rocprofv3 --kernel-trace CSV of bench.py: per (kernel, grid) launches per step, average and
per-step time, per-stream busy time and the busy union -- steps are delimited by a kernel that is launched exactly ONCE per
step (the fused AdamW kernel; --step-kernel to name another).  (Through round 3 the delimiter was "six dense attention-forward
launches"; since the CLS-only last layer there are five per step, and the per-step columns of profiles/r03{d,e,f}_trace_summary.json
were 6/5 too large -- profiles/r03f_trace_summary_corrected.json.)

    python tools/trace_summary.py gpurun_out/prof/x_kernel_trace.csv [--top 40] [--stream main|all] [--seq]
"""
import argparse
import collections
import csv
import json
import re


def short(n):
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void at::native::", "")
    return n[:56]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--stream", default="main")
    ap.add_argument("--seq", action="store_true", help="print the main-stream launch sequence of one step")
    ap.add_argument("--step-kernel", default="adamw_kernel", help="substring of a kernel launched once per step (its END closes a step)")
    ap.add_argument("--json", default="")
    ap.add_argument("--skip-last", type=int, default=0, help="ignore the last K steps (bench.py's eager probe steps)")
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.csv)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [int(r["End_Timestamp"]) + 1 for r in rows if a.step_kernel in r["Kernel_Name"]]      # a step = (end of one AdamW, end of the next]
    if len(starts) < 4:
        raise SystemExit(f"fewer than 4 launches of a kernel matching {a.step_kernel!r}: cannot delimit steps")
    if a.skip_last:
        starts = starts[:-a.skip_last]
    lo, hi = starts[len(starts) // 2], starts[-1]
    n = len(starts) - 1 - len(starts) // 2
    sel = [r for r in rows if lo <= int(r["Start_Timestamp"]) < hi]
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    by = collections.defaultdict(float)
    for r in sel:
        by[r["Stream_Id"]] += dur(r)
    main_s = max(by, key=by.get)
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel)
    u, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s > ce:
            u += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    u += ce - cs
    summary = {"steps": n, "step_delimiter": a.step_kernel, "wall_us_per_step": (hi - lo) / 1e3 / n, "busy_union_us_per_step": u / 1e3 / n,
               "stream_busy_us_per_step": {s: v / n for s, v in by.items()}, "kernels_per_step": len(sel) / n}
    print(json.dumps(summary))
    pick = [r for r in sel if a.stream == "all" or r["Stream_Id"] == main_s]
    g = collections.defaultdict(lambda: [0, 0.0])
    for r in pick:
        k = (short(r["Kernel_Name"]), r["Grid_Size_X"])
        g[k][0] += 1
        g[k][1] += dur(r)
    table = []
    for k, v in sorted(g.items(), key=lambda kv: -kv[1][1])[:a.top]:
        print(f"{k[0]:56s} {k[1]:>9s} n/step={v[0] / n:5.1f} avg={v[1] / v[0]:7.1f} per-step={v[1] / n:7.1f}")
        table.append({"kernel": k[0], "grid": k[1], "per_step": v[0] / n, "avg_us": v[1] / v[0], "us_per_step": v[1] / n})
    if a.json:
        summary["top"] = table
        json.dump(summary, open(a.json, "w"), indent=1)
    if a.seq:
        prev = None
        s0 = starts[len(starts) // 2]
        for r in rows:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            if s0 - 300000 <= s < starts[len(starts) // 2 + 1] and r["Stream_Id"] == main_s:
                print(f"{(s - s0) / 1e3:9.1f} gap={(s - prev) / 1e3 if prev else 0:6.1f} dur={(e - s) / 1e3:7.1f} "
                      f"{short(r['Kernel_Name'])} {r['Grid_Size_X']}")
                prev = e


if __name__ == "__main__":
    main()
