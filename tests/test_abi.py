"""CPU-only: the C-ABI library loads and exports every symbol include/dph.h (the contract) and include/dph_debug.h (tuning, profiling,
pipelining plumbing, test hooks) declare -- no compute calls."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="dph.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dph_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from densephrases_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    contract, debug = _declared_symbols("dph.h"), _declared_symbols("dph_debug.h")
    assert len(contract) >= 20 and len(debug) >= 10
    # the contract header holds no scaffolding: nothing called dph_debug_*, no tuning keys, no twins / CU-range streams / profiling
    assert not [n for n in contract if n.startswith("dph_debug_") or n in ("dph_index_set_tuning", "dph_index_create_twin", "dph_stream_create_cu_range",
                                                                            "dph_profile_enable", "dph_search_prepare_dev", "dph_scan_counters")]
    assert not set(contract) & set(debug)
    names = sorted(contract + debug)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ but not exported by libdph.so"
    assert sorted(_lib.EXPORTED) == names, (set(names) ^ set(_lib.EXPORTED))
    assert _lib.lib.dph_abi_version() == 7


def test_header_is_self_contained_for_a_plain_c_and_a_cpp_consumer():
    """include/dph.h is the drop-in boundary: a host in C (the cgo / JNI / ctypes stubs of INTEGRATION.md bind exactly this) includes
    it and nothing else -- it must compile on its own as C11 and as C++17 (it used size_t without <stddef.h> until round 5)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None or shutil.which("g++") is None:
        pytest.skip("no host compiler")
    for name in ("dph.h", "dph_debug.h"):
        hdr = os.path.join(ROOT, "include", name)
        subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", hdr], check=True)
        subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", hdr], check=True)


def test_plain_c_example_compiles_links_and_refuses_to_run_without_a_gpu(tmp_path):
    """examples/mips_search.c: MIPS.search's hot path (search, get_idxs, window re-score) from plain C11 over include/dph.h -- compiled
    with -Wall -Wextra -Werror and LINKED against the in-tree libdph.so (every entry point it uses resolves); run here it must stop at
    the device check ("no CPU fallback"), on a GPU box it finds the rows its queries were made from (tools/README.md)."""
    import shutil
    import subprocess
    import torch
    import __graft_entry__ as g
    g.build()
    if shutil.which("gcc") is None:
        pytest.skip("no host compiler")
    from densephrases_amd import _lib
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "mips_search")
    subprocess.run(["gcc", "-std=c11", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "mips_search.c"),
                    "-L", libdir, "-ldph", f"-Wl,-rpath,{libdir}", "-Wl,-rpath-link,/opt/rocm/lib", "-L", "/opt/rocm/lib", "-lm", "-o", exe], check=True)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the run itself belongs to the gpu suite")
    r = subprocess.run([exe, "1000", "2"], capture_output=True, text=True)
    assert r.returncode == 1 and "no GPU" in r.stderr, (r.returncode, r.stdout, r.stderr)


def test_argument_errors_do_not_need_a_gpu():
    from densephrases_amd import _lib
    rc = _lib.lib.dph_index_create(0, -1, 0, None)
    assert rc == -1 and b"bad arguments" in _lib.lib.dph_last_error()
    assert _lib.lib.dph_index_ntotal(None) == 0
    assert _lib.lib.dph_index_destroy(None) == 0


def test_synthetic_generator_statistics():
    from densephrases_amd.synth import synthetic_rows
    a = synthetic_rows(0, 4096, seed=42)
    b = synthetic_rows(1000, 16, seed=42)
    assert a.dtype == np.int8 and a.shape == (4096, 768)
    np.testing.assert_array_equal(a[1000:1016], b)                   # depends on the global row index only
    assert abs(a.mean() - 40.0) < 0.1 and abs(a.std() - 12.0) < 0.1  # float_to_int8(N(0,0.6^2)) = 40 + 12 z
    assert not np.array_equal(a, synthetic_rows(0, 4096, seed=43))


def test_host_logic_split_sentences_and_normalize():
    from densephrases_amd.index import MIPS, normalize_answer, split_sentences
    s = split_sentences("One two. Three!  Four? five")          # (the second blank is a token: it starts the sentence)
    assert [t for t, _ in s] == ["One two.", "Three!", " Four?", "five"]
    assert [p for _, p in s] == [0, 9, 16, 23]
    assert normalize_answer("The  Quick, brown fox!") == "quick brown fox"
    each = {"context": "aa bb. [PAR] cc dd ee. [PAR] ff", "start_pos": 16, "end_pos": 18}
    out = MIPS.adjust(dict(each))
    assert out["context"] == "cc dd ee." and out["start_pos"] == 3 and out["end_pos"] == 5


def test_scan_kernel_isa_audit():
    """The scan kernels (flat / masked / unit scans with and without the aux k-step, the PQ coarse filter scan) own a[140:255] by hand
    (aux operands, probe masks / queue atomic, staging buffers of the HBM feed): the compiler must never touch that range outside the
    asm statements, must not insert a vmcnt wait of its own into the streaming loop, and the kernels must not spill."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_audit", os.path.join(ROOT, "tools", "audit_scan_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.audit(verbose=False) == 0
    # ... and the hot kernels of the coarse quantizer / the PQ scans keep everything in registers (no scratch, no spills)
    assert mod.audit_no_scratch(verbose=False) == 0


def test_scan_work_queue_segments_partition_the_tiles():
    """The guided self-scheduling of the flat scan's work queue (the function the kernel calls, through its host twin):
    the segments of u = 0, 1, 2, ... tile [0, n_tiles) exactly once, lengths never grow, never fall below seg_min
    except for the last one, and an empty answer stays empty for every later u."""
    import ctypes as C
    from densephrases_amd import _lib
    f = _lib.lib.dph_debug_guided_segment
    for n_tiles, grid, seg in [(5_312_500, 256, 64), (5_312_500, 256, 16), (166_016, 256, 64), (256, 256, 1), (1, 256, 1),
                               (0, 256, 4), (1000, 7, 3), (20752, 256, 20), (664_063, 304, 64)]:
        pos, prev, n_seg, u = 0, None, 0, 0
        while True:
            ln = C.c_int64(0)
            first = f(u, n_tiles, grid, seg, C.byref(ln))
            if first >= n_tiles:
                break
            assert first == pos, (n_tiles, grid, seg, u)
            assert ln.value >= 1 and (ln.value >= seg or first + ln.value == n_tiles)
            assert prev is None or ln.value <= prev
            pos, prev, n_seg, u = first + ln.value, ln.value, n_seg + 1, u + 1
        assert pos == n_tiles
        for later in (u + 1, u + 2 * grid, u + 10 * grid):
            assert f(later, n_tiles, grid, seg, C.byref(C.c_int64(0))) >= n_tiles
        if n_tiles >= 64 * grid * seg:
            assert n_seg <= 40 * grid            # ~20 pops per workgroup on a big shard, not n_tiles / seg
    assert f(-1, 10, 256, 1, None) == -1 and f(0, 10, 0, 1, None) == -1


def test_fused_scan_visits_exactly_the_tiles_the_ladder_level_skipped():
    """The full scan behind a fused finest ladder level (tuning key ladder_fuse) must read every tile that is not a multiple of
    the level's stride exactly once, in ascending order -- through the host twin of the plan and of the kernel's index
    arithmetic (a multiply and a shift instead of a division: exact only under the guard the plan applies)."""
    import ctypes as C
    from densephrases_amd import _lib
    f = _lib.lib.dph_debug_fused_tile
    visit = C.c_int64()

    def tile(n, s, v):
        return int(f(n, s, v, C.byref(visit)))

    rng = np.random.default_rng(3)
    for s in (2, 3, 4, 7, 8, 16, 32, 33, 64, 128, 1000, 1024):
        # small shards: every visit
        for n in (4 * s, 4 * s + 1, 5 * s - 1, 5 * s, 5 * s + 1, 997, 4096, 5000):
            first = tile(n, s, 0)
            if n <= 4 * s:
                assert first == -1 and visit.value == 0                  # a shard of a few tiles is not fused
                continue
            want = [t for t in range(n) if t % s]
            assert visit.value == len(want) == n - (n + s - 1) // s
            assert [tile(n, s, v) for v in range(len(want))] == want
            assert tile(n, s, len(want)) == -1 and tile(n, s, -1) == -1
        # full-size shards (170 M and 375 M rows of 32-row tiles, and odd sizes): ends, period boundaries, random visits
        for n in (5_312_500, 11_718_750, 11_718_751, 7_654_321):
            tile(n, s, 0)
            nv = visit.value
            if nv == 0:
                continue                                                 # the plan declined: the scan is simply not fused
            assert nv == n - (n + s - 1) // s
            vs = np.unique(np.concatenate([np.arange(0, 2000), np.arange(nv - 2000, nv), rng.integers(0, nv, 20000),
                                           (np.arange(1, 2000)[:, None] * (s - 1) + np.array([-1, 0, 1])).ravel() % nv]))
            got = np.array([tile(n, s, int(v)) for v in vs])
            q = vs // (s - 1)
            np.testing.assert_array_equal(got, vs + q + 1)               # the v-th tile that is not a multiple of s
            assert (got % s != 0).all() and got.max() < n
    # the strides the ladder uses at full size must actually be fused (a declined plan would silently cost 1/S of a scan)
    for s in (16, 32, 64):
        tile(5_312_500, s, 0)
        assert visit.value > 0, s


@pytest.mark.gpu
def test_plain_c_example_finds_the_rows_its_queries_were_made_from(tmp_path):
    """examples/mips_search.c on the GPU: 2 M synthetic rows, 4 queries (each a stored row de-quantised again) -- search, get_idxs
    and the window re-score through the C-ABI from a process that is not python."""
    import shutil
    import subprocess
    import __graft_entry__ as g
    g.build()
    if shutil.which("gcc") is None:
        pytest.skip("no host compiler")
    from densephrases_amd import _lib
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "mips_search")
    subprocess.run(["gcc", "-std=c11", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "mips_search.c"),
                    "-L", libdir, "-ldph", f"-Wl,-rpath,{libdir}", "-Wl,-rpath-link,/opt/rocm/lib", "-L", "/opt/rocm/lib", "-lm", "-o", exe], check=True)
    r = subprocess.run([exe, "2000000", "4"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "every query found the row it was made from" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert "8 query rows" in r.stdout and " 0 uncertified" in r.stdout, r.stdout


def test_every_entry_point_refuses_null_and_zero_arguments_without_crashing():
    """No exception crosses the C ABI and no argument is trusted (`return int status + dph_last_error()`): every exported
    function called with NULL for every pointer / handle and 0 for every number returns -- an error code (negative) with a message, or a
    benign value (`dph_index_destroy(NULL)` = 0 like free(NULL), `dph_index_dim(NULL)` = 768, ...) -- and the process survives.  In a
    child process: a segfault would otherwise take the test run with it; the child names the call it is about to make."""
    code = r'''
import ctypes as C, sys
from densephrases_amd import _lib
for name in _lib.EXPORTED:
    fn = getattr(_lib.lib, name)
    args = [None if (t in (C.c_void_p, C.c_char_p) or hasattr(t, "contents")) else (0.0 if t in (C.c_float, C.c_double) else 0) for t in fn.argtypes]
    print("CALL", name, flush=True)
    rc = fn(*args)
    if isinstance(rc, int) and rc < 0:
        assert _lib.lib.dph_last_error(), name + ": an error code without a message"
    print("DONE", name, rc if not isinstance(rc, bytes) else "", flush=True)
print("ALL", len(_lib.EXPORTED), flush=True)
'''
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT)
    lines = r.stdout.strip().splitlines()
    assert r.returncode == 0 and lines and lines[-1].startswith("ALL"), (r.returncode, lines[-2:], r.stderr[-1500:])
    done = {ln.split()[1]: ln.split()[2:] for ln in lines if ln.startswith("DONE")}
    benign = {"dph_abi_version", "dph_last_error", "dph_device_count", "dph_index_destroy", "dph_index_ntotal", "dph_index_dim",
              "dph_index_rows_dev", "dph_host_free_pinned", "dph_index_stored_rows", "dph_stream_destroy"}
    for name, rc in done.items():
        if name not in benign:
            assert rc and int(rc[0]) < 0, (name, rc, "accepted NULL / zero arguments")
