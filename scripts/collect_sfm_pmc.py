#!/usr/bin/env python3
"""Per-kernel sums of the SQ counter passes of `bench.py --config sfm3` (scripts/_prof_sfm_pmc.sh): all launches of a kernel
together, with the derived figures that say what binds it - waves per SIMD while it runs, VALU-busy fraction, cycles per
VALU instruction, the wait fractions.  usage: collect_sfm_pmc.py <dirA> <dirB> <out.json>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

SIMDS, XCDS = 1024, 8
acc = defaultdict(lambda: defaultdict(float))
launches = defaultdict(set)
for d in sys.argv[1:3]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cvhip::", "").strip()
            if "ransac" not in name and "search" not in name and "match" not in name:
                continue
            acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
            if d == sys.argv[1]:
                launches[name].add(r["Dispatch_Id"])
res = {"note": "sums over all launches of one `bench.py --config sfm3 --steps 1 --warmup 1` run (2 steps); SQ_WAVE_CYCLES / SQ_WAIT_* / "
               "SQ_ACTIVE_* in quad-cycles summed over waves; GRBM_GUI_ACTIVE summed over the 8 XCDs",
       "git_head": os.environ.get("CVHIP_GIT_HEAD"), "kernels": {}}
for name, c in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0)):
    k = dict(c)
    k["launches"] = len(launches[name])
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / XCDS
    if gui and c.get("SQ_WAVES"):
        k["busy_cycles_M"] = round(gui / 1e6, 3)
        k["waves_per_simd_while_running"] = round(c["SQ_WAVE_CYCLES"] * 4.0 / SIMDS / gui, 2)
        k["valu_busy"] = round(c["SQ_ACTIVE_INST_VALU"] * 4.0 / SIMDS / gui, 3)
        k["valu_instr_per_wave"] = round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1)
        k["wave_cycles_per_valu_instr"] = round(c["SQ_WAVE_CYCLES"] * 4.0 / max(c["SQ_INSTS_VALU"], 1.0), 2)
        k["wait_any_frac"] = round(c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1.0), 3)
        k["wait_inst_any_frac"] = round(c["SQ_WAIT_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1.0), 3)
    res["kernels"][name] = k
json.dump(res, open(sys.argv[3], "w"), indent=1)
for name, k in res["kernels"].items():
    if "busy_cycles_M" in k:
        print(f"{name[:60]:60s} n={k['launches']:4d} busy {k['busy_cycles_M']:8.2f} Mcyc  waves/SIMD {k['waves_per_simd_while_running']:5.2f}  valu_busy {k['valu_busy']:5.3f}  "
              f"VALU/wave {k['valu_instr_per_wave']:9.1f}  cyc/VALU {k['wave_cycles_per_valu_instr']:6.2f}  wait_any {k['wait_any_frac']:.2f} wait_inst {k['wait_inst_any_frac']:.2f} "
              f"scratch wr/rd {k.get('SQ_INSTS_VMEM_WR', 0):.3g}/{k.get('SQ_INSTS_VMEM_RD', 0):.3g}")
