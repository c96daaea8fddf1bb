#!/usr/bin/env python3
"""Text timeline of one find_ransac call from a rocprofv3 --kernel-trace CSV: every kernel between the n-th
ransac_coord_max_kernel and the following ransac_inlier_mask_kernel, start / end in us relative to the first, with its queue.
usage: ransac_gantt.py <dir> [n]"""
import csv
import glob
import sys

rows = []
for f in glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
starts = [i for i, r in enumerate(rows) if "ransac_coord_max_kernel" in r["Kernel_Name"]]
i0 = starts[n]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cvhip::", "")
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:9.1f} {e:9.1f} {e - s:8.1f}  q{r.get('Queue_Id', '?'):>3s}  {name[:60]}")
    if "ransac_coord_max_kernel" in name and s > 0:
        break
