#!/usr/bin/env python3
"""Text timeline of one dense step from a rocprofv3 --kernel-trace CSV: the kernels between the n-th and the (n+1)-th
64x64-level window_stats launch.   usage: dense_gantt.py <dir> [n]"""
import csv
import glob
import sys

rows = []
for f in glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
firsts = [i for i, r in enumerate(rows) if "search2_filter_split_kernel<false, 1>" in r["Kernel_Name"]]
i0, i1 = firsts[n] - 1, firsts[n + 1] - 1
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cvhip::", "")
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:9.1f} {e:9.1f} {e - s:8.1f}  q{r.get('Queue_Id', '?'):>3s}  grid {r.get('Grid_Size', '?'):>9s}  {name[:50]}")
