#!/usr/bin/env python3
"""Turn two rocprofv3 PMC runs of `bench.py --steps 1 --warmup 1 --no-cpu-baseline`
(one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE, as MI355X_MICROARCH.md prescribes:
separate passes) into profiles/<tag>_traffic.json.

gfx950 corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports half the bytes of coalesced
streaming reads — confirmed here on expand_grid_kernel, whose 8 B/cell read of 16.7 M cells shows
65 550 KiB instead of 131 072 — so reads are doubled; WRITE_SIZE is exact (window_stats_kernel:
262 144 KiB for 16 B x 16.7 M px).  Units are KiB."""
import csv
import glob
import json
import sys
from collections import defaultdict


def per_kernel(d, counter):
    rows = list(csv.DictReader(open(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0])))
    acc = defaultdict(lambda: [0.0, 0])
    for r in rows:
        if r["Counter_Name"] != counter:
            continue
        # (template arguments kept: the timed and the candidate-counting instantiation of the box kernel are different kernels;
        # the profiled run holds identical steps only - bench.py --no-count-step)
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cvhip::", "").strip()
        acc[name][0] += float(r["Counter_Value"])
        acc[name][1] += 1
    return acc


fdir, wdir, out, steps_in_run = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
F, W = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
res = {"note": "KiB summed over all launches of one bench step; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024",
       "kernels": {}}
for name in sorted(F):
    f, n = F[name]
    w = W.get(name, [0.0, 1])[0]
    res["kernels"][name] = {"launches_per_step": n / steps_in_run, "fetch_size_kib_per_step": f / steps_in_run,
                            "write_size_kib_per_step": w / steps_in_run,
                            "hbm_bytes_per_step": (2 * f + w) * 1024 / steps_in_run}
# provenance: the commit the profile was taken at (handed in, the GPU box has no .git) and the dense kernels' sources
import hashlib
import os
from pathlib import Path

_root = Path(__file__).resolve().parent.parent
_h = hashlib.sha256()
for _rel in ("cybervision_amd/csrc/corr_kernels.hip", "cybervision_amd/csrc/box_body.inc"):
    _h.update((_root / _rel).read_bytes())
res["kernel_source_sha16"] = _h.hexdigest()[:16]
res["git_head"] = os.environ.get("CVHIP_GIT_HEAD")
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res["kernels"].items() if k.startswith("search3_box")}, indent=1))
