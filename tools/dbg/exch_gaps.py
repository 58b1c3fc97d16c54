"""Idle time of the main queue in front of every bottleneck exchange of one replayed step (rocprofv3 kernel trace)."""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
k = len(idx) - 3
step = rows[idx[k - 1] + 1:idx[k] + 1]
mq = collections.Counter(r["Queue_Id"] for r in step).most_common(1)[0][0]
q = [r for r in step if r["Queue_Id"] == mq]
tot = 0.0
for a, b in zip(q, q[1:]):
    if "exchange" in b["Kernel_Name"]:
        g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
        tot += g
        print(f"  {g:7.1f} us before {b['Kernel_Name'][:40]}")
print("total idle before exchanges %.0f us; step wall %.0f us" % (tot, (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3))
