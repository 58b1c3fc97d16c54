"""One replayed (or eager) training step out of a rocprofv3 kernel trace: the kernels of every queue in start order
with their durations, plus totals by kernel family.  Steps are cut at the AdamW kernel.

    python tools/step_seq.py gpurun_out/<run>/x_kernel_trace.csv [--step K] [--window T0 T1] [--families]
"""
import argparse
import collections
import csv
import re


def short(n):
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n).replace("void at::native::", "at::").replace("(anonymous namespace)::", "")
    return n[:72]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--step", type=int, default=-3)
    ap.add_argument("--window", type=float, nargs=2, default=None)
    ap.add_argument("--families", action="store_true")
    ap.add_argument("--all-queues", action="store_true", help="with --families: one table per queue, not only the busiest")
    a = ap.parse_args()
    rows = sorted(csv.DictReader(open(a.csv)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
    k = a.step if a.step >= 0 else len(idx) + a.step
    step = rows[idx[k - 1] + 1:idx[k] + 1]
    t0 = int(step[0]["Start_Timestamp"])
    print("kernels", len(step), "wall us", (int(step[-1]["End_Timestamp"]) - t0) / 1e3, "queues",
          dict(collections.Counter(r["Queue_Id"] for r in step)))
    if a.families:
      qs = [q for q, _ in collections.Counter(r["Queue_Id"] for r in step).most_common()]
      for main_q in (qs if a.all_queues else qs[:1]):
        tot, cnt = collections.Counter(), collections.Counter()
        for r in step:
            if r["Queue_Id"] != main_q:
                continue
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            n = r["Kernel_Name"]
            if "at::native" in n or "rocclr" in n:
                key = "torch small (<10us)" if d < 10 else "torch: " + short(n)[:48]
            elif n.startswith("Cijk"):
                key = "hipblaslt"
            else:
                key = short(n).split("I")[0].split("(")[0][:24]
            tot[key] += d
            cnt[key] += 1
        print("main queue" if main_q == qs[0] else "queue", main_q, "kernel time us", round(sum(tot.values()), 1))
        for key, v in tot.most_common(40):
            print(f"{v:9.1f} us n={cnt[key]:4d}  {key}")
      return
    for r in step:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        if a.window and not (a.window[0] <= s <= a.window[1]):
            continue
        print(f"q{r['Queue_Id']} {s:8.0f} ({e - s:6.1f}) g={r['Grid_Size_X']:>8} {short(r['Kernel_Name'])}")


if __name__ == "__main__":
    main()
