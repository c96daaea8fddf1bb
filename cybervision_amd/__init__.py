"""cybervision_amd — MI355X (gfx950) backend for the hot path of zlogic/cybervision.

The product is the C-ABI shared library ``libcvhip.so`` (include/cvhip.h): hand-written HIP
kernels for dense stereo correlation, ORB extraction, keypoint matching and RANSAC scoring,
behind the reference's own GPU-backend boundary.  This package is the thin ctypes front-end
used by tests and bench.py plus the host-side mirror of the reference interface
(`PointCorrelations`, `orb.extract_points`, `FundamentalMatrix`).  There is NO CPU fallback:
if the extension is missing or no GPU is present, calls raise.
"""
from . import synth  # noqa: F401

__all__ = ["synth"]
