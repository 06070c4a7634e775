"""CPU-only: the C-ABI library loads and exports every symbol include/dph.h declares (no compute calls)."""
import ctypes
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "dph.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dph_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from densephrases_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in dph.h but not exported by libdph.so"
    assert sorted(_lib.EXPORTED) == names, (set(names) ^ set(_lib.EXPORTED))
    assert _lib.lib.dph_abi_version() == 3


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
    """The scan kernel owns a[160:255] by hand (staging buffers of the HBM feed): the compiler must never touch that
    range outside the asm statements, and the kernel must not spill."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_audit", os.path.join(ROOT, "tools", "audit_scan_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.audit(verbose=False) == 0


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
