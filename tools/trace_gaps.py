"""Per-kernel durations and the gaps between consecutive kernels of one trust-region iteration, from a rocprofv3
--kernel-trace csv (graph replay of bench.py).  python tools/trace_gaps.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("ssba::", "").replace("void ", "")))
rows.sort()
# the last 600 kernels: steady-state graph replays
rows = rows[-1200:]
dur = defaultdict(list)
gap = defaultdict(list)
for i, (s, e, n) in enumerate(rows):
    dur[n].append(e - s)
    if i:
        gap[n].append(s - rows[i - 1][1])
tot_d = sum(sum(v) for v in dur.values())
tot_g = sum(sum(v) for v in gap.values())
print(f"kernels {len(rows)}  sum durations {tot_d/1e3:.1f} us  sum gaps {tot_g/1e3:.1f} us")
for n in sorted(dur, key=lambda k: -sum(dur[k])):
    d, g = dur[n], gap[n]
    print(f"{n[:60]:60s} n={len(d):4d} avg {sum(d)/len(d)/1e3:7.2f} us  gap-before avg {sum(g)/max(len(g),1)/1e3:6.2f} us  share {100*sum(d)/tot_d:5.1f}%")
