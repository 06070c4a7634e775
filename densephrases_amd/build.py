"""Build libdph.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["dph_api.hip", "dph_scan.hip", "dph_quant.hip", "dph_refine.hip", "dph_select.hip", "dph_window.hip",
           "dph_ivf.hip", "dph_build.hip", "dph_kmeans.hip", "dph_pq.hip"]
HEADERS = [os.path.join(CSRC, "dph_internal.h"), os.path.join(HERE, "..", "include", "dph.h"), os.path.join(HERE, "..", "include", "dph_debug.h")]
OUT = os.path.join(CSRC, "libdph.so")
HOST_SRC = os.path.join(CSRC, "dph_host.cpp")           # the C++ host half of MIPS.search_phrase (pybind11, g++)
# ... and the tables it includes (tools/gen_sentencizer_tables.py writes them from densephrases_amd/sentencizer.py / unicodedata)
HOST_DEPS = [HOST_SRC, os.path.join(CSRC, "dph_sentencizer_tables.inc"), os.path.join(CSRC, "dph_unicode_punct.inc")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
# the scan pops its work queue one segment ahead: the atomic's result must stay in flight, not be read back at once by
# the wave-reduction form the atomic optimizer would give it (dph_scan.hip, "queue_pop")
EXTRA_FLAGS = {"dph_scan.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]}


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def host_ext_path() -> str:
    import sysconfig
    return os.path.join(HERE, "_dph_host" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def needs_build() -> bool:
    return _stale(OUT, [os.path.join(CSRC, s) for s in SOURCES] + HEADERS) or _stale(host_ext_path(), HOST_DEPS)


def _build_host(force: bool, verbose: bool) -> None:
    out = host_ext_path()
    if not force and not _stale(out, HOST_DEPS):
        return
    import sysconfig
    import pybind11
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-pthread", "-shared", "-fPIC", "-I" + sysconfig.get_paths()["include"],
           "-I" + pybind11.get_include(), HOST_SRC, "-o", out + f".tmp{os.getpid()}"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(out + f".tmp{os.getpid()}", out)



def build(force: bool = False, verbose: bool = True) -> str:
    """Idempotent and safe under concurrent callers (bench.py runs one process per GPU): an exclusive file lock around
    the check-and-build, objects compiled in parallel (one hipcc per source), the library written under a temporary
    name and renamed into place."""
    import fcntl
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not needs_build():
            return OUT
        _build_host(force, verbose)
        if not force and not _stale(OUT, [os.path.join(CSRC, s) for s in SOURCES] + HEADERS):
            return OUT
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

        def compile_one(src: str) -> str:
            obj = os.path.join(CSRC, src[:-4] + ".o")
            if force or _stale(obj, [os.path.join(CSRC, src)] + HEADERS):
                cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", src, "-o", obj]
                if verbose:
                    print(" ".join(cmd), file=sys.stderr)
                subprocess.run(cmd, cwd=CSRC, check=True)
            return obj

        with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
            objs = list(ex.map(compile_one, SOURCES))
        tmp = OUT + f".tmp{os.getpid()}"
        subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs, cwd=CSRC, check=True)
        os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
