#!/usr/bin/env python3
"""Turn a rocprofv3 --pmc SQ pass of `bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-count-step` into
profiles/<tag>_pmc.json: per kernel INSTANTIATION (template arguments kept: `search3_box_kernel<false, false, false>` is the
timed one, `<true, ...>` the candidate-counting one - they are different kernels), summed over the launches of ONE bench
step.  The profiled run must hold identical steps only (--no-count-step: timed step + instrumented step = 2).

    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
              GRBM_GUI_ACTIVE -d <dir> --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    collect_pmc.py <dir> <out.json> <steps_in_run>

Derived per kernel: valu_busy = SQ_ACTIVE_INST_VALU * 4 / (SIMDs * GRBM_GUI_ACTIVE / XCDs)  - SQ_ACTIVE_* count
quad-cycles summed over all SIMDs, GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, PMC section)."""
import csv
import glob
import json
import sys
from collections import defaultdict

SIMDS, XCDS = 1024, 8
d, out, steps_in_run = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = []
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
acc = defaultdict(lambda: defaultdict(float))
launches = defaultdict(set)
per_dispatch = defaultdict(lambda: defaultdict(float))
grid = {}
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cvhip::", "").strip()
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    launches[name].add(r["Dispatch_Id"])
    per_dispatch[(name, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    grid[(name, r["Dispatch_Id"])] = int(r["Grid_Size"])
res = {"note": "per bench step; SQ_* summed over all waves / SIMDs, quad-cycle units for ACTIVE/WAIT; GRBM_GUI_ACTIVE summed over 8 XCDs",
       "kernels": {}}
for name, c in sorted(acc.items()):
    k = {cn: v / steps_in_run for cn, v in c.items()}
    k["launches_per_step"] = len(launches[name]) / steps_in_run
    g = k.get("GRBM_GUI_ACTIVE", 0.0)
    if g and "SQ_ACTIVE_INST_VALU" in k:
        k["valu_busy"] = round(k["SQ_ACTIVE_INST_VALU"] * 4.0 / (SIMDS * g / XCDS), 4)
    if k.get("SQ_WAVES"):
        k["valu_instr_per_wave"] = round(k.get("SQ_INSTS_VALU", 0.0) / k["SQ_WAVES"], 1)
    # the largest launch on its own (the last one of that size: the uninstrumented instantiation): on the small
    # launches the GPU is mostly idle, so utilisation figures are quoted for this one
    big = max((key for key in per_dispatch if key[0] == name), key=lambda key: (grid[key], int(key[1])))
    b = dict(per_dispatch[big])
    if b.get("GRBM_GUI_ACTIVE") and "SQ_ACTIVE_INST_VALU" in b:
        b["valu_busy"] = round(b["SQ_ACTIVE_INST_VALU"] * 4.0 / (SIMDS * b["GRBM_GUI_ACTIVE"] / XCDS), 4)
        b["cycles_per_valu_instr"] = round(b["SQ_ACTIVE_INST_VALU"] * 4.0 / max(b.get("SQ_INSTS_VALU", 1.0), 1.0), 3)
    b["grid"] = grid[big]
    k["largest_launch"] = b
    res["kernels"][name] = k
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
for n in sorted(res["kernels"]):
    if n.split("<")[0] in ("search3_box_kernel", "window_stats_kernel", "search_range_kernel", "cross_check_kernel"):
        k = res["kernels"][n]
        print(n, {x: k.get(x) for x in ("launches_per_step", "valu_busy", "valu_instr_per_wave", "SQ_INSTS_VALU", "SQ_WAVES")})
