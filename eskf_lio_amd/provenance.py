"""What a measurement was taken ON: hashes that tie a committed profile (profiles/*_summary.json) to the library a
later bench.py run loads.  `library_sha256` is the built libvgicp_hip.so byte for byte; `kernel_source_sha256` covers
the sources and build flags that decide the registration kernels' code (a rebuild of the host side alone, or on
another box, changes the first but not the second)."""
from __future__ import annotations

import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
KERNEL_SOURCES = ("csrc/vgicp_kernels.hip", "csrc/vgicp_device.h", "csrc/vgicp_device_fn.h", "csrc/vgicp_math.h",
                  "csrc/Makefile")


def _sha(paths) -> str:
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def library_sha256(path: str | None = None) -> str | None:
    path = path or os.path.join(_HERE, "lib", "libvgicp_hip.so")
    return _sha([path]) if os.path.exists(path) else None


def kernel_source_sha256() -> str | None:
    paths = [os.path.join(_HERE, p) for p in KERNEL_SOURCES]
    return _sha(paths) if all(os.path.exists(p) for p in paths) else None
