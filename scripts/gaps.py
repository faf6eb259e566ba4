"""usage: gaps.py <rocprofv3 output dir>  -- GPU busy/idle and per-kernel gap statistics from a kernel trace"""
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]            # second half: the timed region
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
busy = 0; last_end = t0; gaps = defaultdict(list); dur = defaultdict(list)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-50:]
    if s > last_end:
        gaps[name].append(s - last_end)
    busy += max(0, e - max(s, last_end)); last_end = max(last_end, e)
    dur[name].append(e - s)
span = t1 - t0
print(f"span {span/1e6:.2f} ms busy {busy/1e6:.2f} ms idle {1-busy/span:.3f}")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:16]:
    g = gaps.get(k, [0])
    print(f"{k:52s} n={len(v):5d} avg_us={sum(v)/len(v)/1e3:8.1f} tot_ms={sum(v)/1e6:7.2f}  gap_before avg_us={sum(g)/max(1,len(g))/1e3:7.1f} n={len(g)}")
