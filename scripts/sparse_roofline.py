#!/usr/bin/env python3
"""Per-kernel roofline table of config 5 (`bench.py --config sfm3`) from the committed profile summaries:
profiles/<tag>_sfm3_kernel_stats.csv (rocprofv3 --kernel-trace --stats, N steps), <tag>_sfm3_traffic.json (FETCH_SIZE /
WRITE_SIZE passes, gfx950 corrections applied by collect_traffic.py) and <tag>_sfm3_pmc.json (SQ pass).
    sparse_roofline.py <tag> <steps in the stats run> > profiles/<tag>_sfm3_roofline.md
    sparse_roofline.py <tag> <steps> dense > profiles/<tag>_dense_roofline.md      (the headline run's summaries)"""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

tag, steps = sys.argv[1], int(sys.argv[2])
dense = len(sys.argv) > 3 and sys.argv[3] == "dense"  # the headline run (bench.py, 4096^2 pair) instead of config 5
mid = "" if dense else "_sfm3"
prof = Path(__file__).resolve().parent.parent / "profiles"
traffic = json.load(open(prof / f"{tag}{mid}_traffic.json"))["kernels"]
pmc = json.load(open(prof / f"{tag}{mid}_pmc.json"))["kernels"]
HBM_PEAK = 8.0e12

# what binds each kernel, with the evidence column that shows it
BOUND = {
    "brief_kernel": "HBM (gathers of f64 blur samples)", "blur_h_kernel": "HBM (streaming)", "blur_v_kernel": "HBM (streaming)",
    "moments_kernel": "HBM (patch gathers)", "expand_grid_kernel": "HBM (streaming)", "contrast_kernel": "HBM (streaming)",
    "fast_score_kernel": "VALU (FAST-9 bisection)", "harris_kernel": "f64 VALU (7x7 taps)", "minmax_kernel": "latency (one pass, atomics)",
    "match_kernel": "VALU popcount + LDS broadcast", "ransac_count_kernel": "packed-f32 VALU",
    "ransac_perspective_lm_kernel": "latency: the longest LM loop of the round (f64, one wave per SIMD)",
    "ransac_perspective_root_kernel": "f64 VALU / HBM (288 B pencil per sample)", "ransac_perspective_pencil_kernel": "f64 VALU (QR + cubic)",
    "ransac_tied_sum_kernel": "latency: serial f64 sum (the reference's order)", "ransac_refit_kernel": "latency: serial dot-product chains, one workgroup",
    "ransac_pick_best_kernel": "latency (one workgroup)", "window_stats_kernel": "VALU (serial f32 chain)", "search_range_kernel": "latency + VALU",
    "search3_box_single_kernel": "VALU issue (dot4 + DPP)", "search3_box_kernel": "VALU issue (dot4 + DPP)", "search2_filter_kernel": "dot4 issue (candidate filter: first pass, steep pair)",
    "cross_check_kernel": "latency (dependent loads)", "search3_fallback_kernel": "latency (persistent, mostly empty lists)",
}
# the headline bench's first (untimed) step counts candidates: its search kernels are the COUNT = true instantiations,
# several times slower - they are left out and the search kernels averaged over the other steps
COUNTING = ("search3_box_kernel", "search3_box_single_kernel", "search2_filter_kernel", "search3_fallback_kernel")
acc = defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(prof / f"{tag}{mid}_kernel_stats.csv")):
    full = r["Name"].split("(")[0].replace("void ", "").replace("cvhip::", "")
    name = full.split("<")[0]
    if dense and name in COUNTING and "<true" in full:
        continue
    acc[name][0] += int(r["Calls"])
    acc[name][1] += int(r["TotalDurationNs"])
what = "Headline (4096^2 pair, 7 levels)" if dense else "Config 5 (3 x 2048^2 perspective views)"
print(f"# {what}, per-kernel device time and traffic per step ({tag})\n")
print("Sources: `{0}{2}_kernel_stats.csv` ({1} steps), `{0}{2}_traffic.json`, `{0}{2}_pmc.json`; HBM peak 8 TB/s; "
      "VALU busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) over ALL launches of the kernel in a step "
      "(small levels included, and for the headline run the counting step's instantiations too: the full-resolution launch of "
      "the box kernel alone is at 1.0, see `bench.py`'s roofline).\n".format(tag, steps, mid))
print("| kernel | launches/step | ms/step | HBM MB/step | TB/s | of HBM peak | VALU busy | bound by |")
print("|---|---|---|---|---|---|---|---|")
total = 0.0
for name, (calls, ns) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if name not in BOUND:
        continue
    n_steps = steps - 1 if dense and name in COUNTING else steps
    ms = ns / 1e6 / n_steps
    calls = calls * steps / n_steps
    total += ms
    by = traffic.get(name, {}).get("hbm_bytes_per_step")
    tbs = by / (ms / 1e3) / 1e12 if by else None
    vb = pmc.get(name, {}).get("valu_busy")
    print(f"| `{name}` | {calls / steps:.0f} | {ms:.3f} | {by / 1e6:.1f} | {tbs:.2f} | {tbs * 1e12 / HBM_PEAK:.3f} | "
          f"{vb:.2f} | {BOUND[name]} |" if by and vb is not None else f"| `{name}` | {calls / steps:.0f} | {ms:.3f} | - | - | - | - | {BOUND[name]} |")
print(f"\nSum of the listed kernels: {total:.1f} ms per step" + ("." if dense else " (kernels of different streams overlap: the RANSAC "
      "generator runs under the counting kernels)."))
