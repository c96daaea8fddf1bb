#!/usr/bin/env python3
"""bench.py — headline benchmark: Mpixels/s of dense stereo correlation on a synthetic
4096x4096 grayscale pair (reference-default 11x11 window, affine parameter set, 7 pyramid
levels; BASELINE.json configs[2]/[3]) on N MI355X.

One step = one complete multi-level dense correlation of the pair: for every level
{forward search, reverse search, cross-check forward, cross-check reverse}
(PointCorrelations::correlate_images, correlation/mod.rs:217-245, driven as in
reconstruction.rs:554-588) plus complete() into a device-resident full-resolution grid.
Inputs (both u8 pyramids) are resident in HBM before the timed region.

N > 1: the SAME pair is row-sharded over the ranks (strong scaling), bands are all-gathered
with RCCL after every sharded search pass; every rank ends with the full, identical grid.

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` for the
dominant kernel (search3_box_kernel, timed live with HIP events on its own stream) and
`cpu_baseline` (the C restatement of the reference's --mode=cpu path, oracle/, timed on this
node's host cores on a bounded sample).  At N = 1 the same line also carries, measured after the
timed region: `readback` (SURVEY 8(d)'s t_dense with the final grid landing in page-locked HOST
memory - per pair, and with the transfer of one pair under the search of the next),
`geometry_sweep` (the same 4096^2 workload with the epipolar lines tilted by 3 .. 90 degrees) and
`sfm3` (BASELINE config 5, per-stage times).

`python bench.py --gpus N` without a launcher starts the N ranks itself (torch.distributed.run as a
child process, before anything touches the GPU) and exits with its code.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# v_dot4_u32_u8 issues at 4 cycles per wave-instruction per SIMD on gfx950 (measured with
# scripts/micro/valu_rate.hip: 4.4 at 8 waves/SIMD), i.e. 64 lanes x 4 MAC / 4 cycles per SIMD:
DOT4_PEAK_TMACS = 256 * 4 * 64 * 4 / 4 * 2.4e9 / 1e12   # = 157.3 T multiply-adds/s, the filter kernel's roof


RESULT_BANDS = 0                 # cvhip_ctx_set_result_bands in the host-destination modes: 0 = the library's choice by size (the
                                 # binding's setting; 6 bands at 4096^2, scripts/result_bands_probe.py)
SEARCH_KERNEL = "search3_box_kernel"   # the kernel class "search" times (search version 3, the default)
SEARCH_KERNEL_INST = "search3_box_kernel<false, false, false>"   # the instantiation the timed region launches (no candidate counter)


def algorithmic_work(level_dims, candidates):
    """Algorithmic bytes / multiply-adds of ALL search-kernel launches of one step.  Per (level,
    direction) pass, each array touched once: searched pixel = img1 u8 (1) + statistics word (8) + search interval (4)
    + result cell written (8) = 21 B; target pixel = img2 u8 (1) + statistics word (8) = 9 B -> 30 B per level pixel.
    Multiply-adds: 121 per evaluated candidate (the reference's 11x11 window)."""
    b = 0
    for (w1, h1, w2, h2) in level_dims:
        n1, n2 = w1 * h1, w2 * h2
        b += n1 * 21 + n2 * 9   # forward
        b += n2 * 21 + n1 * 9   # reverse
    return b, 121.0 * candidates


def survey_bytes(level_dims):
    """~40 B per level pixel per direction for the WHOLE step (every kernel), each array once."""
    return sum(40 * (w1 * h1 + w2 * h2) for (w1, h1, w2, h2) in level_dims)


def _profile_json(name):
    f = ROOT / "profiles" / name
    return json.loads(f.read_text()) if f.exists() else None


KERNEL_SOURCES = ("cybervision_amd/csrc/corr_kernels.hip", "cybervision_amd/csrc/box_body.inc")


def kernel_source_sha16():
    """sha256 of the dense kernels' sources: the PMC profiles record it when they are collected, so the line can say
    whether the instruction counts it divides by still describe the kernels that ran."""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        h.update((ROOT / rel).read_bytes())
    return h.hexdigest()[:16]


def profile_source(name):
    d = _profile_json(name)
    if not d:
        return None
    return {"profile": f"profiles/{name}", "collected_at_commit": d.get("git_head"), "kernel_source_sha16": d.get("kernel_source_sha16"),
            "matches_the_kernels_that_ran": d.get("kernel_source_sha16") == kernel_source_sha16()}


def traffic_per_launch(world):
    """HBM bytes per search-kernel launch from the rocprofv3 PMC passes of this same command
    (FETCH_SIZE and WRITE_SIZE in separate passes, gfx950 corrections applied: scripts/collect_traffic.py);
    committed under profiles/.  None when no PMC data matches this configuration."""
    d = _profile_json("current_traffic.json")
    if world != 1 or not d:
        return None
    k = d["kernels"].get(SEARCH_KERNEL_INST)
    return round(k["hbm_bytes_per_step"] / k["launches_per_step"]) if k else None


def step_traffic(world):
    """Measured HBM bytes of ALL kernels of one step (same PMC passes)."""
    d = _profile_json("current_traffic.json")
    if world != 1 or not d:
        return None
    return round(sum(v["hbm_bytes_per_step"] for n, v in d["kernels"].items() if "kernel" in n and "::" not in n and not n.startswith("__")))


def valu_profile(world):
    """VALU instruction counts of the search kernel from the SQ PMC pass of this same command
    (scripts/collect_pmc.py -> profiles/current_pmc.json): wave-instructions per step, and for the full-resolution
    launch the pipe-busy fraction SQ_ACTIVE_INST_VALU * 4 / SIMDs / (GRBM_GUI_ACTIVE / XCDs)."""
    d = _profile_json("current_pmc.json")
    if world != 1 or not d or SEARCH_KERNEL_INST not in d["kernels"]:
        return None
    k = d["kernels"][SEARCH_KERNEL_INST]
    big = k.get("largest_launch", {})
    return {"wave_instr_per_step": k.get("SQ_INSTS_VALU"), "valu_busy": big.get("valu_busy"),
            "cycles_per_valu_instr": big.get("cycles_per_valu_instr")}


def measure_sfm3(size, steps, warmup, dev=None, pencil=0, listener=True, lanczos=False):
    """BASELINE config 5 ("3-image perspective SFM: ORB + RANSAC F-matrix on GPU + pairwise dense correlation"):
    three synthetic size^2 perspective views, resident in HBM as u8 pyramids; one step = per-level ORB on the three
    images, 3 x matcher (threshold 48), 3 x perspective find_ransac (device RANSAC + LM refit), 3 x dense correlation
    with the perspective parameter set (reconstruction.rs:261-277, 400-526, 540-588).  Single GPU ("replicas only":
    the sparse stage does not shard)."""
    import torch

    from cybervision_amd import correlation, fundamentalmatrix, reconstruction, synth
    from cybervision_amd.fundamentalmatrix import ProjectionMode

    views, K, poses = synth.make_sfm_views(size)
    lsteps = synth.optimal_scale_steps(size, size)
    torch.cuda.set_device(0)
    own_dev = dev is None  # (the headline run hands its handle over: one set of side streams per process, see Device::aux)
    if own_dev:
        dev = correlation.create_gpu_context(ordinal=0, stream=torch.cuda.current_stream().cuda_stream)
    pyr = [[torch.from_numpy(l).cuda() for l in synth.box_pyramid(v, lsteps)] for v in views]
    # (resident, padded, complete before the timed region: the dense stage uses the levels in place)
    pyr = [reconstruction.padded_pyramid(p)[0] for p in pyr]
    # lanczos: the reference's own pyramids instead - Lanczos3 resizes of the full-resolution views on the device, inside the
    # step, rebuilt per stage as reconstruction.rs:146-162, 421-422, 567-568 does (a different workload: other level images)
    images = [torch.from_numpy(v).cuda() for v in views] if lanczos else None
    # the 7-point pencil: 0 = rows 5 / 6 of the thin SVD as the reference writes it (the library's default), 1 = null space
    fundamentalmatrix.set_pencil(dev, pencil)
    acc, n_pairs, matches, inliers, dense_cells = {}, 0, [], [], []
    t0 = time.perf_counter()
    for it in range(warmup + steps):
        if it == warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            acc = {}
        res = reconstruction.reconstruct_pairs(dev, pyr, ProjectionMode.Perspective, seed=5, borrow=True, listener=listener, images=images)
        for k, v in res["timings_ms"].items():
            acc[k] = acc.get(k, 0.0) + v
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for (i, j), e in res["pairs"].items():
        n_pairs += e["f"] is not None
        matches.append(int(len(e["matches"])))
        inliers.append(int(len(e["inliers"])) if e["inliers"] is not None else 0)
        dense_cells.append(reconstruction.match_count(e["xy"]) if "xy" in e else 0)
    fundamentalmatrix.set_pencil(dev, fundamentalmatrix.PENCIL_THIN_SVD)
    if own_dev:
        dev.close()
    stage_ms = {k: round(v / steps, 3) for k, v in acc.items()}
    return {"pyramids": ("Lanczos3 on the device inside the step (cvhip_resize_lanczos3), rebuilt per stage as the reference does" if lanczos
                         else "2x2 box pyramids, prebuilt and resident: SourceImage::resize (reconstruction.rs:146-162) is EXCLUDED from the step"),
            "pencil": "thin_svd_rows_5_6 (reference)" if pencil == 0 else "null_space (textbook)",
            "ransac_listener": "report_status + report_matches attached (reconstruction.rs:510-518)" if listener else "none",
            "dense_pairs": "the three pairs' correlations side by side, a device handle (stream) each (correlate_dense_set)" if images is None else "one after the other",
            "size": size, "levels": lsteps + 1, "steps": steps, "ms_per_step": round(dt * 1e3 / steps, 3), "stage_ms": stage_ms,
            "dense_mpixels_per_s": round(n_pairs * size * size / 1e6 / (stage_ms["dense"] / 1e3), 2),
            "whole_pipeline_mpixels_per_s": round(3 * size * size / 1e6 / (dt / steps), 2),
            "keypoints": [int(len(k[0])) for k in res["keypoints"]], "matches": matches, "ransac_inliers": inliers,
            "dense_matches": dense_cells, "pairs_with_f": n_pairs}


def bench_sfm3(args):
    """`--config sfm3`: config 5 as a line of its own (a secondary line, not the headline metric)."""
    size = 2048 if args.size == 4096 else args.size
    m = measure_sfm3(size, args.steps, args.warmup, pencil=args.pencil)
    print(json.dumps({
        "metric": f"Mpixels/s dense correlation, 3 x {size}x{size} perspective views (3 pairs), config 5", "secondary": True,
        "value": m["dense_mpixels_per_s"], "unit": "Mpixels/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": m["ms_per_step"], "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8 / f32 / f64 as in the reference", "data": "synthetic",
        "config": {"workload": f"3 perspective views {size}x{size} of one depth surface (synth.make_sfm_views), per-level ORB, "
                               "matcher thr 48, perspective RANSAC (20 x 50 000 samples, early exit) + LM refit, 3 pairwise "
                               f"dense correlations (9 stripes, thr 0.5, {m['levels']} levels)", "parallelism": "single GPU"},
        "stage_ms_per_step": m["stage_ms"], "pencil": m["pencil"],
        "whole_pipeline_mpixels_per_s": m["whole_pipeline_mpixels_per_s"],
        "keypoints": m["keypoints"], "matches": m["matches"], "ransac_inliers": m["ransac_inliers"],
        "dense_matches": m["dense_matches"],
    }), flush=True)


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD torch.distributed.run (one rank per
    GPU over RCCL) - before this process has imported torch or touched the GPU - and return its exit code."""
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=4096, help="image side (default: BASELINE's 4096)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--simulate-shard", default="", help="num/den: time one rank's share of a den-GPU band-mode run on 1 GPU")
    ap.add_argument("--cpu-sample", type=int, default=512, help="side of the first crop timed on the CPU")
    ap.add_argument("--config", default="dense4096", choices=["dense4096", "sfm3"],
                    help="dense4096 (default): the headline metric; sfm3: BASELINE config 5, a secondary line")
    ap.add_argument("--pencil", type=int, default=0, choices=[0, 1], help="--config sfm3: 7-point pencil (0 = the reference's thin-SVD rows, 1 = null space)")
    ap.add_argument("--no-extras", action="store_true", help="skip readback / geometry_sweep / sfm3 in the headline line")
    ap.add_argument("--no-count-step", action="store_true",
                    help="profiling runs (scripts/_run_prof.sh): skip the untimed candidate-counting step, so that every step of the "
                         "run launches the same kernel instantiations; the line then carries no candidate count")
    ap.add_argument("--sweep-tilts", default="3,10,30,45,60,90", help="tilts (degrees) of geometry_sweep")
    args = ap.parse_args()
    if args.config == "sfm3":
        return bench_sfm3(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))  # nothing has touched the GPU yet: the ranks are child processes

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE = {world}")
    # CVHIP_BENCH_BACKEND=gloo: rehearsal of the N-rank code path on a box with ONE GPU - every rank uses cuda:0 and
    # the gather is staged through the host (sharding.make_allgather); the numbers it prints are not a benchmark
    backend = os.environ.get("CVHIP_BENCH_BACKEND", "nccl")
    rehearsal = backend != "nccl"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend, rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from cybervision_amd import correlation, sharding, synth

    W = H = args.size
    steps = synth.optimal_scale_steps(W, H)

    def resident(p):
        # u8 level image in HBM with 64 readable bytes behind it, so that the library can use it in place
        # (cvhip_ctx_set_borrow_inputs) instead of copying it into its own padded buffer on every call
        buf = torch.zeros(p.numel() + 64, dtype=torch.uint8, device="cuda")
        buf[:p.numel()].copy_(p.reshape(-1))
        return buf[:p.numel()].view(p.shape[0], p.shape[1])

    def resident_pyramids(tilt_deg=0.0):
        # the synthetic pair and its 2x2 box pyramids, built on the device (synth.make_pair_torch: the same integer
        # arithmetic as synth.make_pair, byte-identical - tests/test_sharding_cpu.py)
        a, b, _ = synth.make_pair_torch(W, H, tilt_deg=tilt_deg, device="cuda")
        pa, pb = synth.box_pyramid_torch(a, steps), synth.box_pyramid_torch(b, steps)
        return [resident(p) for p in pa], [resident(p) for p in pb]

    d1, d2 = resident_pyramids()
    img1, img2 = d1[0].cpu().numpy(), d2[0].cpu().numpy()  # host copies of the pair, for the CPU baseline
    level_dims = [(int(p.shape[1]), int(p.shape[0]), int(q.shape[1]), int(q.shape[0])) for p, q in zip(d1, d2)]

    stream = torch.cuda.current_stream()
    dev = correlation.create_gpu_context(ordinal=local_rank, stream=stream.cuda_stream)
    pc = correlation.PointCorrelations(dev, (W, H), (W, H), synth.F_HORIZONTAL, correlation.ProjectionMode.Affine)
    pc.set_borrow_inputs(True)
    # the pyramid is complete in HBM before the timed region starts: the library may compute a level's window statistics
    # on its side stream, under the coarse levels' search (cvhip_ctx_set_stats_ahead; same bits)
    pc.set_stats_ahead(True)
    band_mode = False
    final_gather = None
    sim = None
    collective = None
    comm = None

    # N > 1 only: a collective that never completes (a rank died, a link is down) would block every other rank inside
    # RCCL for good - ncclCommInitRank included, so the watchdog starts BEFORE the communicator is made.  It ends this
    # process - loudly, with a non-zero code - when no step or fence has completed for three minutes, so a stuck run
    # fails instead of holding its GPUs until someone else's limit.
    progress = {"t": time.monotonic(), "what": "setup"}

    def beat(what):
        progress["t"] = time.monotonic()
        progress["what"] = what

    if world > 1:
        import threading

        def watchdog():
            while True:
                time.sleep(5.0)
                idle = time.monotonic() - progress["t"]
                if idle > 180.0:
                    print(f"[bench] rank {rank}: no progress for {idle:.0f} s after '{progress['what']}' - aborting", flush=True)
                    os._exit(3)

        threading.Thread(target=watchdog, daemon=True).start()

    if args.simulate_shard:  # single-GPU emulation of ONE rank's share of an N-GPU run (no collectives)
        num, den = (int(v) for v in args.simulate_shard.split("/"))
        sim = (num, den)
        if not pc.set_row_band(num, den):
            raise SystemExit("--simulate-shard needs row-local geometry")
        band_mode = True
    elif world > 1:
        # Collectives: the library's own RCCL path (cvhip_rccl_*: a communicator per device handle, every collective on
        # the handle's stream, no torch in the data path); torch.distributed only carries the 128-byte id.  If that
        # cannot be set up (or in the one-GPU gloo rehearsal, where RCCL refuses several ranks per GPU) the
        # torch.distributed hook is used instead - the line says which.
        if not rehearsal:
            # ncclCommInitRank is a collective: a rank that cannot even load RCCL must not leave the others waiting in
            # it.  So every rank first probes the library locally (making an id loads librccl and resolves the symbols),
            # the ranks agree, and only then rank 0's id goes round and the communicator is created.
            uid_local = None
            try:
                uid_local = sharding.RcclCommunicator.unique_id()
            except Exception as exc:  # noqa: BLE001 - reported, then the hook path takes over
                print(f"[bench] rank {rank}: library RCCL path unavailable ({exc}); using the torch.distributed hook", flush=True)
            usable = torch.tensor([1 if uid_local is not None else 0], device="cuda")
            dist.all_reduce(usable, op=dist.ReduceOp.MIN)
            if int(usable[0]) == 1:
                uid = [uid_local if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                try:
                    beat("cvhip_rccl_create")
                    comm = sharding.RcclCommunicator(dev, uid[0], rank, world)
                    collective = "library RCCL (cvhip_rccl_*), gather to rank 0"
                except Exception as exc:  # noqa: BLE001
                    print(f"[bench] rank {rank}: cvhip_rccl_create failed ({exc}); using the torch.distributed hook", flush=True)
                    comm = None
        ok = torch.tensor([1 if comm is not None else 0], device="cpu" if rehearsal else "cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)      # all ranks take the same path
        if int(ok[0]) == 0 and comm is not None:
            comm.close()
            comm = None
        band_mode = pc.set_row_band(rank, world)   # independent bands + halo, one final gather ...
        if comm is not None:
            if band_mode:
                final_gather = lambda cells, nbytes, n, d: pc.gather_bands_rccl(comm, 0) if d == 0 else None  # noqa: E731 (both planes in one call)
            else:
                pc.set_row_shard_rccl(comm)        # ... or, for non-row-local geometry, an all-gather per sharded pass
        else:
            collective = "torch.distributed all_gather hook"
            gather = sharding.make_allgather(rank, world)
            if band_mode:
                final_gather = gather
            else:
                pc.set_row_shard(rank, world, gather)
    out_xy = torch.empty((H, W, 2), dtype=torch.int32, device="cuda")
    out_corr = torch.empty((H, W), dtype=torch.float32, device="cuda")

    l0_events = []  # (start, end) of the full-resolution level of every timed step (SURVEY §8d secondary metric)

    def step(timed=False):
        beat("step start")
        pc.first_pass = True
        for i in range(steps + 1):
            k = steps - i
            if timed and k == 0:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
            pc.correlate_images(d1[k], d2[k], 1.0 / float(1 << k))
            if timed and k == 0:
                e1.record(stream)
                l0_events.append((e0, e1))
        if final_gather is not None:  # the single RCCL gather: forward bands of the full-resolution level
            g = pc.level_grid(correlation.CorrelationDirection.Forward)
            final_gather(g["cells"], g["rows_per_shard"] * g["lw"] * 4, world, 0)   # match plane
            final_gather(g["scores"], g["rows_per_shard"] * g["lw"] * 4, world, 2)  # score plane
        pc.complete(out_xy=out_xy, out_corr=out_corr)

    def fence():
        beat("fence start")
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # one counting pass (untimed): candidates evaluated per step, needed for the flop count
    cand_local = 0
    if not args.no_count_step:
        pc.set_profiling(False, True)
        step()
        cand_local = pc.get_counters()["candidates"]
        pc.set_profiling(False, False)
    for _ in range(max(args.warmup - (0 if args.no_count_step else 1), 0)):
        step()
    fence()
    # The timed region: K steps.  HIP events are live inside it - they bracket every launch of the dominant
    # (search) kernel and the full-resolution level - but only in its LAST step: with any timing event in flight
    # this runtime profiles every dispatch of the step (0.45 ms per step, however few events), so instrumenting
    # all K steps would make the measurement the largest "kernel" after the search.
    pc.set_profiling(0, False)
    t0 = time.perf_counter()
    for i in range(args.steps):
        last = i == args.steps - 1
        if last:
            pc.set_profiling(2, False)
        step(timed=last)
    fence()
    dt = time.perf_counter() - t0
    l0_ms = sum(a.elapsed_time(b) for a, b in l0_events) / max(len(l0_events), 1)
    ktimes_last = pc.get_kernel_times()
    search_ms_local = ktimes_last["search"]["ms"]  # the box-kernel launches of ONE step (the first pass runs the
    search_launches = ktimes_last["search"]["launches"]  # candidate filter: class "search_filter", not counted here)
    # per-class breakdown of one more (untimed) step
    pc.set_profiling(1, False)
    step()
    fence()
    ktimes = pc.get_kernel_times()
    pc.set_profiling(0, False)

    extras = world == 1 and sim is None and not args.no_extras
    readback = geometry_sweep = sfm3 = boundary_path = None
    if extras:
        # ---- t_dense INCLUDING the final readback of the forward grid into host memory
        # (GpuContext::complete_process lands in a host Grid, gpu/mod.rs:210-216).  Page-locked destinations.
        # (a) per pair: complete() returns when the 201 MB are in host memory, then the next pair starts;
        # (b) pipelined, as a reconstruction correlates pair after pair (reconstruction.rs:680-730): the transfer of
        #     pair i runs on the handle's copy stream under the search of pair i + 1 (cvhip_ctx_set_async_readback,
        #     two staging sets, two host buffers); the clock stops when the last grid is in host memory.
        host = [(torch.empty((H, W, 2), dtype=torch.int32).pin_memory(), torch.empty((H, W), dtype=torch.float32).pin_memory())
                for _ in range(2)]

        def step_host(i):
            pc.first_pass = True
            for j in range(steps + 1):
                k = steps - j
                pc.correlate_images(d1[k], d2[k], 1.0 / float(1 << k))
            hx, hc = host[i & 1]
            pc.complete(out_xy=hx.numpy(), out_corr=hc.numpy())

        rb_steps = max(3, min(args.steps, 10))
        step_host(0)
        step_host(1)
        fence()
        t1 = time.perf_counter()
        for i in range(rb_steps):
            step_host(i)
        dev.synchronize()
        rb_sync_ms = (time.perf_counter() - t1) * 1e3 / rb_steps
        same_host = bool(torch.equal(host[(rb_steps - 1) & 1][0], out_xy.cpu()))
        # (c) per pair as (a), the last level in result bands (cvhip_ctx_set_result_bands): band b on its way to the host
        #     while the bands behind it are searched
        pc.set_result_bands(RESULT_BANDS)
        step_host(0)
        step_host(1)
        fence()
        t1 = time.perf_counter()
        for i in range(rb_steps):
            step_host(i)
        dev.synchronize()
        rb_band_ms = (time.perf_counter() - t1) * 1e3 / rb_steps
        live_bands = pc.result_bands()
        same_host = same_host and bool(torch.equal(host[(rb_steps - 1) & 1][0], out_xy.cpu()))
        pc.set_result_bands(1)
        pc.set_async_readback(True)
        step_host(0)
        dev.synchronize()
        t1 = time.perf_counter()
        for i in range(rb_steps):
            step_host(i)
        dev.synchronize()
        rb_pipe_ms = (time.perf_counter() - t1) * 1e3 / rb_steps
        pc.set_async_readback(False)
        same_host = same_host and bool(torch.equal(host[(rb_steps - 1) & 1][0], out_xy.cpu()))
        mpx_ = W * H / 1e6
        readback = {"t_dense_with_readback_ms": round(rb_sync_ms, 4), "mpixels_per_s_with_readback": round(mpx_ / (rb_sync_ms / 1e3), 1),
                    "t_dense_with_readback_banded_ms": round(rb_band_ms, 4), "result_bands": live_bands,
                    "mpixels_per_s_with_readback_banded": round(mpx_ / (rb_band_ms / 1e3), 1),
                    "t_dense_with_readback_pipelined_ms": round(rb_pipe_ms, 4),
                    "mpixels_per_s_with_readback_pipelined": round(mpx_ / (rb_pipe_ms / 1e3), 1),
                    "bytes_to_host_per_pair": W * H * 12, "steps": rb_steps, "host_grid_equals_device_grid": same_host,
                    "note": "complete() into page-locked host memory (12 B/px: int32 x, y + f32 score); banded = the last level in "
                            "row bands, each copied out under the search of the bands behind it (cvhip_ctx_set_result_bands); "
                            "pipelined = the transfer of pair i under the search of pair i+1 (cvhip_ctx_set_async_readback)"}
        del host

        # ---- The reference's real call modes.  PointCorrelations::correlate_images issues FOUR
        # backend calls per level (correlation/mod.rs:217-245: correlate_images fwd, rev with the images exchanged,
        # cross_check_filter fwd, rev) with HOST level images (`Grid<u8>`, gpu/mod.rs:218-274), and complete_process lands
        # in host memory (gpu/mod.rs:210-216) - the sequence INTEGRATION.md binds.  Timed here, per pair, clock stopped when
        # the forward grid is in host memory (pageable numpy arrays on both sides, like Rust's Vec):
        #   four_call_host    the four calls per level on host images, complete() to host
        #   level_call_host   cvhip_correlate_level (the optional fused call) on host images, complete() to host
        #   four_call_device  the four calls per level on the HBM-resident pyramid, grid left in HBM (the headline's
        #                     workload through the reference's call sequence)
        hp1, hp2 = [p.cpu().numpy() for p in d1], [p.cpu().numpy() for p in d2]
        hxy, hcorr = np.empty((H, W, 2), dtype=np.int32), np.empty((H, W), dtype=np.float32)
        pcb = correlation.PointCorrelations(dev, (W, H), (W, H), synth.F_HORIZONTAL, correlation.ProjectionMode.Affine)
        pcb.set_fuse_level_calls(True)  # (what the binding's GpuContext::new does: INTEGRATION.md)

        hcells = np.empty((H, W), dtype=np.uint32)

        def pair(p1, p2, fused, to_host):
            pcb.first_pass = True
            for j in range(steps + 1):
                k = steps - j
                pcb.correlate_images(p1[k], p2[k], 1.0 / float(1 << k), fused=fused)
            if to_host == "packed":
                pcb.complete_packed(out_cells=hcells, out_corr=hcorr)
            elif to_host:
                pcb.complete(out_xy=hxy, out_corr=hcorr)
            else:
                pcb.complete(out_xy=out_xy, out_corr=out_corr)

        def timed_pairs(p1, p2, fused, to_host, n=5):
            pair(p1, p2, fused, to_host)
            fence()
            t1 = time.perf_counter()
            for _ in range(n):
                pair(p1, p2, fused, to_host)
            fence()
            return (time.perf_counter() - t1) * 1e3 / n

        want_xy = out_xy.cpu().numpy()
        ms_4h_flat = timed_pairs(hp1, hp2, False, True)   # (round 3's figure: the whole grid copied out behind the last level)
        same_b = bool((hxy == want_xy).all())
        pcb.set_result_bands(RESULT_BANDS)  # (the binding's setting: the grid always goes to the host there)
        ms_4h = timed_pairs(hp1, hp2, False, True)
        live_b = pcb.result_bands()
        same_b = same_b and bool((hxy == want_xy).all()) and live_b > 1
        ms_4hp = timed_pairs(hp1, hp2, False, "packed")
        same_b = same_b and bool((correlation.PointCorrelations.unpack_cells(hcells) == want_xy).all())
        ms_lh = timed_pairs(hp1, hp2, True, True)
        same_b = same_b and bool((hxy == want_xy).all())
        pcb.set_result_bands(1)
        pcb.set_borrow_inputs(True)
        pcb.set_stats_ahead(True)
        ms_4d = timed_pairs(d1, d2, False, False)
        same_b = same_b and bool(torch.equal(out_xy.cpu(), torch.from_numpy(want_xy)))
        pcb.close()
        boundary_path = {"four_call_host_ms": round(ms_4h, 3), "four_call_host_mpixels_per_s": round(W * H / 1e6 / (ms_4h / 1e3), 1),
                         "four_call_host_packed_cells_ms": round(ms_4hp, 3), "four_call_host_one_band_ms": round(ms_4h_flat, 3),
                         "result_bands": live_b,
                         "level_call_host_ms": round(ms_lh, 3), "four_call_device_ms": round(ms_4d, 3),
                         "four_call_device_vs_headline": round(ms_4d / (dt * 1e3 / args.steps), 3),
                         "results_equal_headline": same_b,
                         "note": "per level cvhip_correlate_images x2 + cvhip_cross_check_filter x2 (correlation/mod.rs:217-245), then "
                                 "cvhip_complete; host = pageable level images in, pageable grid out (44 MB up, 201 MB down per pair; packed cells: "
                                 "cvhip_complete_packed, 134 MB down), the last level in result bands (one_band: without); "
                                 "device = the headline's resident pyramid and resident grid through the same four calls"}
        del hp1, hp2, hxy, hcorr, hcells

        # the same four-call host path (pageable level images in, pageable grid out, the binding's settings) on pairs that are
        # NOT rectified - what an unmodified pipeline sees on real inputs: the 4096^2 pair tilted by 10 degrees (stepped box
        # launches) and a 2048^2 perspective pair of synth.make_sfm_views with its true F (perspective parameter set)
        def host_four_call(p1, p2, Fm, proj, n=5):
            hh, ww = p1[0].shape
            oxy, oco = np.empty((hh, ww, 2), dtype=np.int32), np.empty((hh, ww), dtype=np.float32)
            q = correlation.PointCorrelations(dev, (ww, hh), (ww, hh), Fm, proj)
            q.set_fuse_level_calls(True)
            q.set_result_bands(RESULT_BANDS)
            ns = len(p1) - 1

            def one():
                q.first_pass = True
                for j in range(ns + 1):
                    k = ns - j
                    q.correlate_images(p1[k], p2[k], 1.0 / float(1 << k), fused=False)
                q.complete(out_xy=oxy, out_corr=oco)

            one()
            fence()
            t1 = time.perf_counter()
            for _ in range(n):
                one()
            fence()
            ms = (time.perf_counter() - t1) * 1e3 / n
            live = q.result_bands()
            q.close()
            return {"ms": round(ms, 3), "mpixels_per_s": round(ww * hh / 1e6 / (ms / 1e3), 1), "size": int(ww), "result_bands": live,
                    "matched_fraction": round(float((oxy[..., 0] >= 0).mean()), 4)}

        t1p, t2p = resident_pyramids(10.0)
        boundary_path["four_call_host_tilt10"] = host_four_call([p.cpu().numpy() for p in t1p], [p.cpu().numpy() for p in t2p],
                                                               synth.f_tilt(10.0), correlation.ProjectionMode.Affine)
        del t1p, t2p
        psize = 2048 if W >= 2048 else W
        views, Kc, poses = synth.make_sfm_views(psize)
        psteps = synth.optimal_scale_steps(psize, psize)
        boundary_path["four_call_host_perspective"] = host_four_call(synth.box_pyramid(views[0], psteps), synth.box_pyramid(views[1], psteps),
                                                                     synth.sfm_true_f(Kc, poses[0], poses[1]), correlation.ProjectionMode.Perspective)
        del views

    pc.close()
    pc = None
    pair_pipeline = None
    if extras and hasattr(dev, "side_handle"):
        # ---- pair after pair, as reconstruct_dense's loop issues them (reconstruction.rs:680-730), on TWO contexts used in turn, a
        # device handle (stream) each: the tail of pair i - its last filter and the expansion of the grid, which cannot fill
        # the chip - and the first, small levels of pair i + 1 run side by side.  Throughput of the sequence, not one pair's
        # latency (that is the headline); same grids.
        handles = [dev, dev.side_handle(0)]
        ctxs, outs = [], []
        for hnd in handles:
            q = correlation.PointCorrelations(hnd, (W, H), (W, H), synth.F_HORIZONTAL, correlation.ProjectionMode.Affine)
            q.set_borrow_inputs(True)
            q.set_stats_ahead(hnd is dev)
            ctxs.append(q)
            outs.append((torch.empty((H, W, 2), dtype=torch.int32, device="cuda"), torch.empty((H, W), dtype=torch.float32, device="cuda")))
        torch.cuda.synchronize()

        def pair_on(n):
            q, (oxy, oco) = ctxs[n & 1], outs[n & 1]
            q.first_pass = True
            for j in range(steps + 1):
                k = steps - j
                q.correlate_images(d1[k], d2[k], 1.0 / float(1 << k))
            q.complete(out_xy=oxy, out_corr=oco)

        def fence_both():
            for hnd in handles:
                hnd.synchronize()

        for n in range(4):
            pair_on(n)
        fence_both()
        n_pp = 20
        t1 = time.perf_counter()
        for n in range(n_pp):
            pair_on(n)
        fence_both()
        pp_ms = (time.perf_counter() - t1) * 1e3 / n_pp
        same_pp = bool(torch.equal(outs[0][0], out_xy) and torch.equal(outs[1][0], out_xy) and
                       torch.equal(outs[1][1].view(torch.int32)[out_xy[..., 0] >= 0], out_corr.view(torch.int32)[out_xy[..., 0] >= 0]))
        pair_pipeline = {"ms_per_pair": round(pp_ms, 4), "mpixels_per_s": round(W * H / 1e6 / (pp_ms / 1e3), 1), "pairs": n_pp,
                         "results_equal_headline": same_pp,
                         "note": "two contexts used in turn, a device handle (stream) each: pair i + 1's first levels under pair i's last filter and expansion"}
        for q in ctxs:
            q.close()
        del ctxs, outs
    if extras:
        # ---- the same workload with the epipolar lines tilted: pairs displaced along the tilted direction
        # (synth.make_pair(tilt_deg=...)), F = synth.f_tilt(theta); everything else as in the headline step
        geometry_sweep = {}
        for tilt in [float(v) for v in args.sweep_tilts.split(",") if v]:
            d1, d2 = None, None
            d1, d2 = resident_pyramids(tilt)
            pcs = correlation.PointCorrelations(dev, (W, H), (W, H), synth.f_tilt(tilt), correlation.ProjectionMode.Affine)
            pcs.set_borrow_inputs(True)

            def step_tilt(level0_events=None):
                pcs.first_pass = True
                for j in range(steps + 1):
                    k = steps - j
                    if level0_events is not None and k == 0:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(stream)
                    pcs.correlate_images(d1[k], d2[k], 1.0 / float(1 << k))
                    if level0_events is not None and k == 0:
                        e1.record(stream)
                        level0_events.append((e0, e1))
                pcs.complete(out_xy=out_xy, out_corr=out_corr)

            step_tilt()
            fence()
            n_sw = 5
            t1 = time.perf_counter()
            for _ in range(n_sw):
                step_tilt()
            fence()
            ms_sw = (time.perf_counter() - t1) * 1e3 / n_sw
            ev = []
            step_tilt(ev)
            fence()
            geometry_sweep[f"{tilt:g}"] = {"ms_per_step": round(ms_sw, 3), "mpixels_per_s": round(W * H / 1e6 / (ms_sw / 1e3), 1),
                                           "level0_ms": round(ev[0][0].elapsed_time(ev[0][1]), 3),
                                           "matched_fraction": round(float((out_xy[..., 0] >= 0).float().mean().item()), 4)}
            pcs.close()
        d1 = d2 = None
        torch.cuda.empty_cache()
        # ---- BASELINE config 5, per-stage times
        # the library's defaults - the reference's own 7-point pencil, both RANSAC thunks attached - and, beside it, the
        # stage times of the textbook pencil and of a call without a listener (what rounds 2-3 quoted)
        sfm_size = 2048 if W == 4096 else max(W // 2, 256)
        sfm3 = measure_sfm3(sfm_size, 5, 1, dev=dev)
        alt = measure_sfm3(sfm_size, 3, 1, dev=dev, pencil=1)
        bare = measure_sfm3(sfm_size, 3, 1, dev=dev, listener=False)
        lz = measure_sfm3(sfm_size, 3, 1, dev=dev, lanczos=True)
        sfm3["other_modes"] = {"null_space_pencil_with_listener": {"ms_per_step": alt["ms_per_step"], "stage_ms": alt["stage_ms"], "ransac_inliers": alt["ransac_inliers"]},
                               "reference_pencil_without_listener": {"ms_per_step": bare["ms_per_step"], "stage_ms": bare["stage_ms"]},
                               "lanczos_pyramids_built_in_the_step": {"ms_per_step": lz["ms_per_step"], "stage_ms": lz["stage_ms"], "pyramids": lz["pyramids"],
                                                                      "ransac_inliers": lz["ransac_inliers"], "dense_matches": lz["dense_matches"]}}

    sharded_ok = None
    if world > 1:
        # the sharded result of the last step against an unsharded run of the same pair in this process (the rank that
        # holds the gathered grid: rank 0 with the library's gather-to-root, every rank with an all-gather)
        pc1 = correlation.PointCorrelations(dev, (W, H), (W, H), synth.F_HORIZONTAL, correlation.ProjectionMode.Affine)
        pc1.set_borrow_inputs(True)
        for i in range(steps + 1):
            k = steps - i
            pc1.correlate_images(d1[k], d2[k], 1.0 / float(1 << k))
        ref_xy, ref_corr = torch.empty_like(out_xy), torch.empty_like(out_corr)
        pc1.complete(out_xy=ref_xy, out_corr=ref_corr)
        fence()
        same = bool(torch.equal(out_xy, ref_xy)) and bool(torch.equal(out_corr.view(torch.int32), ref_corr.view(torch.int32)))
        pc1.close()
        holds_result = rank == 0 or collective is None or "hook" in collective or not band_mode
        if rehearsal:
            print(f"[rehearsal] rank {rank}/{world}: sharded result {'==' if same else '!='} unsharded result", flush=True)
        if holds_result and not same:
            raise SystemExit(f"rank {rank}: sharded result differs from the unsharded one")
        sharded_ok = same
        seen = torch.tensor([1], device="cpu" if rehearsal else "cuda")
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)
        ranks_seen = int(seen[0])

    t = torch.tensor([dt, float(cand_local), search_ms_local], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
        candidates = int(tsum[1])      # bands (+ the small levels every rank computes whole)
        search_ms = float(tmax[2])
    else:
        candidates = int(cand_local)
        search_ms = search_ms_local

    if rank == 0:
        ms_per_step = dt * 1e3 / args.steps
        mpx = W * H / 1e6
        value = mpx / (ms_per_step / 1e3)
        bytes_alg, macs_alg = algorithmic_work(level_dims, candidates)
        search_ms_per_step = search_ms                       # all search-kernel launches of one step (slowest rank)
        launches_per_step = search_launches
        # per-rank share of the algorithmic work when sharded
        ach_gbs = bytes_alg / world / (search_ms_per_step / 1e3) / 1e9
        ach_tmacs = macs_alg / world / (search_ms_per_step / 1e3) / 1e12
        # The dominant kernel is bound by VALU instruction issue, not by HBM and not by the matrix pipe (DESIGN.md
        # section 6).  `achieved` = VALU lane-operations per second of the box kernel: its wave-instructions per step
        # (SQ_INSTS_VALU of the committed PMC pass of this same command - `source` says which, and whether it was
        # collected from the kernels that just ran) x 64 lanes / the duration of ITS launches, measured live with HIP
        # events in the timed region.  `peak` = MI355X_MICROARCH.md's issue rate for plain VALU (a wave64 instruction
        # every 2 cycles per SIMD-32: 256 CU x 4 SIMD x 32 lanes x 2.4 GHz).  The kernel's stream is mostly the
        # half-rate class (v_dot4_u32_u8, DPP-modified adds, 24-bit multiplies, conversions: 4 cycles - `peak_half_rate`),
        # so its pipes are saturated (`valu_busy`, SQ_ACTIVE_INST_VALU against GRBM_GUI_ACTIVE; the two counters run on
        # different clock domains, hence readings slightly above 1) at roughly half of `peak`: `frac` is relative to an
        # ASSUMED issue width of 32 lanes per SIMD and clock, and `frac_of_half_rate_peak` is the same rate against
        # the class the stream is actually made of.  The efficiency figure of the ALGORITHM is
        # `valu_lane_instr_per_candidate` (the reference spends 363 flops per candidate).
        vp = valu_profile(world)
        valu_peak = 256 * 4 * 32 * 2.4e9 / 1e12
        if vp and vp["wave_instr_per_step"]:
            lane_ops = vp["wave_instr_per_step"] * 64.0 / world
            ach_valu = lane_ops / (search_ms_per_step / 1e3) / 1e12
            roofline = {"kernel": SEARCH_KERNEL, "bound": "valu", "achieved": round(ach_valu, 2), "peak": round(valu_peak, 1),
                        "unit": "T lane-ops/s (VALU instructions x 64)", "frac": round(ach_valu / valu_peak, 4),
                        "peak_half_rate": round(valu_peak / 2, 1), "frac_of_half_rate_peak": round(ach_valu / (valu_peak / 2), 4),
                        "valu_busy": vp["valu_busy"], "cycles_per_valu_instr": vp["cycles_per_valu_instr"],
                        "valu_lane_instr_per_candidate": round(vp["wave_instr_per_step"] * 64.0 / max(candidates, 1), 1),
                        "source": profile_source("current_pmc.json")}
        else:  # no PMC data for this configuration: only the HBM figures below are measured
            roofline = {"kernel": SEARCH_KERNEL, "bound": "valu", "achieved": None, "peak": round(valu_peak, 1),
                        "unit": "T lane-ops/s (VALU instructions x 64)", "frac": None}
        roofline.update({
            "traffic": traffic_per_launch(world),
            "traffic_source": profile_source("current_traffic.json"),
            "launches_per_step": launches_per_step,
            "avg_launch_ms": round(search_ms_per_step / max(launches_per_step, 1), 4),
            # secondary: the HBM side of the same kernel (compute-bound by construction: ~30 B per ~72 candidates)
            "hbm": {"achieved": round(ach_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach_gbs / HBM_PEAK_GBS, 5),
                    "algorithmic_bytes_per_step": bytes_alg,
                    "whole_step": {"survey_8d_bytes": survey_bytes(level_dims), "measured_bytes": step_traffic(world),
                                   "achieved_gbs_on_survey_bytes": round(survey_bytes(level_dims) / (ms_per_step / 1e3) / 1e9, 1)}},
            # useful work: the reference-equivalent 121 multiply-adds per candidate against the dot4 issue roof - a rate
            # of useful work, NOT a utilisation (the box filter issues ~14 multiply-adds per candidate)
            "useful_work": {"achieved": round(ach_tmacs, 3), "peak": round(DOT4_PEAK_TMACS, 1), "unit": "T reference-equivalent multiply-adds/s",
                            "ratio": round(ach_tmacs / DOT4_PEAK_TMACS, 4), "algorithmic_macs_per_step": macs_alg},
        })
        result = {
            "metric": "Mpixels/s dense correlation, 4096x4096 pair" if W == 4096 else f"Mpixels/s dense correlation, {W}x{H} pair",
            "value": round(value, 3),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u8 (exact integer filter) + f32/f64 (reference-order exact re-evaluation)",
            "data": "synthetic",
            "config": {"workload": f"{W}x{H} u8 value-noise pair, integer disparity field |d|<={max(W // 64, 1)}px, "
                                   f"affine parameter set (11x11 window, 5 stripes, thr 0.6), F = horizontal epipolar "
                                   f"lines, {steps + 1} pyramid levels (2x2 box), fwd+rev search + 2 cross-checks (the last level: 1 - its reverse filter only feeds the grid complete() drops) per "
                                   f"level + complete() to HBM",
                       "parallelism": ("single GPU" if world == 1 else
                                       (f"row bands x{world} + halo, no exchange between levels, one RCCL all-gather of the final grid"
                                        if band_mode else f"row-sharded x{world}, RCCL all-gather per sharded pass"))
                                      + (f" [EMULATION of shard {sim[0]}/{sim[1]} on one GPU, no collective]" if sim else ""),
                       "candidates_per_step": candidates},
            # SURVEY 8(d) defines t_dense WITH the final readback of the forward grid (page-locked host destination, the last
            # level in result bands); `value` is the HBM -> HBM rate the task's bench contract asks for (inputs and grid resident)
            "value_survey_8d": readback["mpixels_per_s_with_readback_banded"] if readback else None,
            "roofline": roofline,
            # every kernel class, from one extra fully instrumented step after the timed region
            "kernel_ms_per_step": {k: round(v["ms"], 4) for k, v in ktimes.items()},
            # the full-resolution level alone (non-first pass: the dominant level), this rank's share of the rows
            "level0": {"ms": round(l0_ms, 4), "mpixels_per_s": round(mpx / world / (l0_ms / 1e3), 1) if l0_ms > 0 else None},
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import cvref  # CPU oracle: only as the reported baseline, never on the product path

            cores = os.cpu_count() or 1
            cvref.build()

            def run_cpu(S):
                c1 = np.ascontiguousarray(img1[:S, :S])
                c2 = np.ascontiguousarray(img2[:S, :S])
                csteps = synth.optimal_scale_steps(S, S)
                cp1, cp2 = synth.box_pyramid(c1, csteps), synth.box_pyramid(c2, csteps)
                tc = time.perf_counter()
                cvref.correlate_dense(cp1, cp2, synth.F_HORIZONTAL, 0, cores)
                return time.perf_counter() - tc, csteps

            # bounded sample: calibrate on a small crop, then time the largest crop predicted to need <= ~30 s
            S = min(args.cpu_sample, W)
            tc, csteps = run_cpu(S)
            while S * 2 <= W and tc * 4.0 <= 30.0:
                S *= 2
                tc, csteps = run_cpu(S)
            result["cpu_baseline"] = {
                "value": round(S * S / 1e6 / tc, 4), "unit": "Mpixels/s", "cores": cores, "kind": "port",
                "sample": (f"top-left {S}x{S} crop of the same pair" if S < W else f"the whole {S}x{S} pair")
                          + f", full {csteps + 1}-level pyramid, C restatement of --mode=cpu (oracle/), {tc:.2f} s",
            }
        if readback is not None:
            result["readback"] = readback
        if pair_pipeline is not None:
            result["pair_pipeline"] = pair_pipeline
        if geometry_sweep is not None:
            result["geometry_sweep"] = geometry_sweep
        if sfm3 is not None:
            result["sfm3"] = sfm3
        if boundary_path is not None:
            result["boundary_path"] = boundary_path
        if world > 1:
            result["collective"] = collective
            result["ranks_seen"] = ranks_seen
            result["sharded_equals_unsharded"] = sharded_ok
        if rehearsal:
            result["rehearsal"] = f"{backend}: all {world} ranks on one GPU, gather staged through the host - not a measurement"
        print(json.dumps(result), flush=True)

    if pc is not None:
        pc.close()
    if comm is not None:
        comm.close()   # before the device handle it was made on (the library would defer the handle's release otherwise)
    dev.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
