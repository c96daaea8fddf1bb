#!/usr/bin/env python3
"""Timeline of the RANSAC rounds from a rocprofv3 --kernel-trace CSV: per kernel the mean duration, and for the counting
kernel the period between consecutive launches (what a round costs) and the gap before each launch.
usage: ransac_timeline.py <dir with *kernel_trace.csv>"""
import csv
import glob
import sys
from collections import defaultdict

rows = []
for f in glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cvhip::", "")
    by[n].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for n, v in sorted(by.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    if "ransac" in n or "tri_scan" in n:
        d = [e - s for s, e in v]
        print(f"{n[:44]:44s} n={len(v):4d} mean {sum(d) / len(d) / 1e3:8.1f} us  max {max(d) / 1e3:8.1f}")
cnt = by.get("ransac_count_kernel", [])
if len(cnt) > 2:
    periods = [cnt[i + 1][0] - cnt[i][0] for i in range(len(cnt) - 1)]
    gaps = [cnt[i + 1][0] - cnt[i][1] for i in range(len(cnt) - 1)]
    periods = [p for p in periods if p < 5e6]
    gaps = [g for g in gaps if g < 5e6]
    print(f"count kernel: period mean {sum(periods) / len(periods) / 1e3:.1f} us, gap to the previous count mean {sum(gaps) / len(gaps) / 1e3:.1f} us")
    print("first 24 (duration, gap after) us:", [(round((e - s) / 1e3), round(g / 1e3)) for (s, e), g in zip(cnt[:24], gaps[:24])])
lmk = by.get("ransac_perspective_lm_kernel", [])
if lmk:
    print("lm first 12 durations us:", [round((e - s) / 1e3) for s, e in lmk[:12]])
