"""Build libcvhip.so (the HIP extension) in-tree with hipcc for gfx950.

`python -m cybervision_amd.build` or `cybervision_amd.build.build()`.  hipcc cross-compiles
without a GPU.  -ffp-contract=off is REQUIRED for parity: the reference (Rust) never fuses
a*b+c, and hipcc's default for device code is -ffp-contract=fast.

Every source is compiled to its own object (in parallel, only where it or a header changed) and
the objects are linked into the one shared library; no relocatable device code is needed because
no kernel calls across translation units.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OBJ = PKG / "build"
LIB = PKG / "libcvhip.so"
SOURCES = ["cvhip_api.hip", "corr_kernels.hip", "orb_kernels.hip", "ransac_kernels.hip", "track_kernels.hip", "resize_kernels.hip", "cvhip_rccl.hip"]
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-Wall", "-Wno-unused-function",
]
# per source.  ransac_kernels.hip: no SLP vectorisation - it packs the counting kernel's adjacent f32 fmas into
# v_pk_fma_f32, which holds a SIMD for ~8.8 cycles against 4 + 4 for the two plain instructions.
# (The dense kernels' hand-written v_pk_add_f32 / v_pk_mul_f32 pairs are the other way round: written out as plain
# instructions window_stats_kernel ran 0.54 -> 0.66 ms.)
# -amdgpu-mfma-vgpr-form: the counting screen's f32 MFMAs (ransac_count_mfma_kernel) write their results where the vector
# instructions that decide on them read them - without it every accumulator is copied out of the AGPRs (8 v_accvgpr_read per chunk)
EXTRA_FLAGS = {"ransac_kernels.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]}
LINK = ["-ldl", "-pthread"]  # cvhip_rccl.hip opens librccl.so.1 lazily (dlopen): no link-time dependency on RCCL


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need ROCm under /opt/rocm)")


def _headers() -> list[Path]:
    return [*CSRC.glob("*.hpp"), *CSRC.glob("*.inc"), *CSRC.glob("*.h"), PKG.parent / "include" / "cvhip.h", Path(__file__)]


def _stale(target: Path, deps: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def needs_build() -> bool:
    return _stale(LIB, [CSRC / s for s in SOURCES] + _headers())


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return LIB
    OBJ.mkdir(exist_ok=True)
    cc, hdrs = hipcc(), _headers()

    def compile_one(src: str) -> Path:
        obj = OBJ / (src + ".o")
        if force or _stale(obj, [CSRC / src] + hdrs):
            # (CVHIP_EXTRA_FLAGS: e.g. -DCVHIP_ABLATIONS for scripts/box_ablation*.sh - build with --force, and again without)
            cmd = [cc, *FLAGS, *EXTRA_FLAGS.get(src, []), *os.environ.get("CVHIP_EXTRA_FLAGS", "").split(), "-c", str(CSRC / src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *[str(o) for o in objs], *LINK]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
