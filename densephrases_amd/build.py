"""Build libdph.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["dph_api.hip", "dph_scan.hip", "dph_select.hip", "dph_window.hip", "dph_ivf.hip"]
OUT = os.path.join(CSRC, "libdph.so")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "dph_internal.h"),
                                                       os.path.join(HERE, "..", "include", "dph.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    """Idempotent and safe under concurrent callers (bench.py runs one process per GPU): an exclusive file lock around
    the check-and-build, output written under a temporary name and renamed into place."""
    import fcntl
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not needs_build():
            return OUT
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        tmp = OUT + f".tmp{os.getpid()}"
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", tmp] + SOURCES
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, cwd=CSRC, check=True)
        os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
