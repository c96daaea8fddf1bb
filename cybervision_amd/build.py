"""Build libcvhip.so (the HIP extension) in-tree with hipcc for gfx950.

`python -m cybervision_amd.build` or `cybervision_amd.build.build()`.  hipcc cross-compiles
without a GPU.  -ffp-contract=off is REQUIRED for parity: the reference (Rust) never fuses
a*b+c, and hipcc's default for device code is -ffp-contract=fast.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libcvhip.so"
SOURCES = ["cvhip_api.hip", "corr_kernels.hip", "orb_kernels.hip", "ransac_kernels.hip", "track_kernels.hip", "resize_kernels.hip", "cvhip_rccl.hip"]
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-Wall", "-Wno-unused-function",
]
LINK = ["-ldl", "-pthread"]  # cvhip_rccl.hip opens librccl.so.1 lazily (dlopen): no link-time dependency on RCCL


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need ROCm under /opt/rocm)")


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = list(CSRC.glob("*")) + [PKG.parent / "include" / "cvhip.h", Path(__file__)]
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return LIB
    cmd = [hipcc(), *FLAGS, "-o", str(LIB), *[str(CSRC / s) for s in SOURCES], *LINK]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
