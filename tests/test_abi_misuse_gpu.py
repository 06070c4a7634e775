"""Misuse of a LIVE handle through the raw C ABI (run with -m gpu): calls out of order, sizes out of range, NULL buffers.  Contract
(include/dph.h, SURVEY 8b "Errors"): every entry point returns a negative DPH_E_* code and leaves a message in dph_last_error() --
nothing is written, nothing crashes -- and the handle goes on answering correctly afterwards.  (FAISS raises C++ exceptions that SWIG
turns into RuntimeError, index.py:285-288 catches exactly that around reconstruct.)"""
import ctypes as C

import numpy as np
import pytest

from oracle import mips_oracle as O

pytestmark = pytest.mark.gpu


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _refused(rc, lib, what):
    assert rc < 0, f"{what}: accepted (rc {rc})"
    assert lib.dph_last_error(), f"{what}: an error code without a message"


def test_flat_handle_refuses_misuse_and_keeps_working():
    from densephrases_amd._lib import lib
    rng = np.random.default_rng(3)
    n = 5000
    xb = O.float_to_int8(rng.standard_normal((n, 768), dtype=np.float32) * np.float32(0.6))
    x = rng.normal(0, 0.5, (4, 768)).astype(np.float32)
    D = np.full((4, 10), 7.0, np.float32)
    I = np.full((4, 10), 7, np.int64)
    h = C.c_void_p()
    _refused(lib.dph_index_create(0, -1, 0, C.byref(h)), lib, "create with a negative row count")
    _refused(lib.dph_index_create(9999, n, 0, C.byref(h)), lib, "create on a device that does not exist")
    assert lib.dph_index_create(0, n, 100, C.byref(h)) == 0 and h.value
    _refused(lib.dph_search(h, _p(x), 4, 10, _p(D), _p(I)), lib, "search before finalize")
    _refused(lib.dph_index_upload_rows(h, n - 10, 11, _p(xb)), lib, "upload past the last row")
    _refused(lib.dph_index_upload_rows(h, -1, 5, _p(xb)), lib, "upload at a negative row")
    _refused(lib.dph_index_upload_rows(h, 0, 5, None), lib, "upload from NULL")
    assert lib.dph_index_upload_rows(h, 0, n, _p(xb)) == 0
    assert lib.dph_index_finalize(h, None) == 0
    for what, args in (("k = 0", (h, _p(x), 4, 0, _p(D), _p(I))), ("k = 1025", (h, _p(x), 4, 1025, _p(D), _p(I))),
                       ("a negative number of queries", (h, _p(x), -1, 10, _p(D), _p(I))), ("NULL queries", (h, None, 4, 10, _p(D), _p(I))),
                       ("NULL scores", (h, _p(x), 4, 10, None, _p(I))), ("NULL ids", (h, _p(x), 4, 10, _p(D), None))):
        _refused(lib.dph_search(*args), lib, "search with " + what)
    assert (D == 7.0).all() and (I == 7).all()                          # nothing was written by a refused call
    _refused(lib.dph_search_dev(h, None, 4, 10, None, None, None, None), lib, "device search with NULL buffers")
    out = np.zeros(768, np.float32)
    for bad in (-1, 99, 100 + n, 1 << 40):
        _refused(lib.dph_reconstruct(h, bad, _p(out)), lib, f"reconstruct id {bad}")
    _refused(lib.dph_reconstruct(h, 100, None), lib, "reconstruct into NULL")
    v = np.asarray([5], np.int32)
    _refused(lib.dph_index_set_tuning(h, b"no_such_key", _p(v), 1), lib, "unknown tuning key")
    _refused(lib.dph_index_set_tuning(h, b"max_qb", _p(np.asarray([99], np.int32)), 1), lib, "tuning value out of range")
    _refused(lib.dph_index_set_tuning(h, None, _p(v), 1), lib, "NULL tuning key")
    _refused(lib.dph_search_ivf(h, _p(x), 4, 10, 8, _p(D), _p(I)), lib, "IVF search on a shard without lists")
    # rows may be replaced after finalize -- and then the shard constants are stale: searching is refused until finalize ran again
    assert lib.dph_index_upload_rows(h, 0, 5, _p(xb)) == 0
    _refused(lib.dph_search(h, _p(x), 4, 10, _p(D), _p(I)), lib, "search after an upload without a new finalize")
    assert lib.dph_index_finalize(h, None) == 0
    # window re-score: k / L out of range, NULL candidates
    ids = np.full(8, 100, np.int64)
    doc = np.zeros(8, np.int32)
    word = np.zeros(8, np.int32)
    first = np.zeros(8, np.float32)
    pred = np.zeros(8, np.int32)
    best = np.zeros(8, np.float64)
    arg = np.zeros(8, np.int32)
    _refused(lib.dph_rescore(h, 0, _p(x), 4, 2, 10, _p(ids), _p(doc), _p(word), _p(first), _p(pred), _p(best), _p(arg), None), lib,
             "re-score without idx2id / f2o")
    _refused(lib.dph_rescore(h, 7, _p(x), 4, 2, 10, _p(ids), _p(doc), _p(word), _p(first), _p(pred), _p(best), _p(arg), None), lib, "re-score direction 7")
    _refused(lib.dph_rescore(h, 0, _p(x), 4, 2, 0, _p(ids), _p(doc), _p(word), _p(first), _p(pred), _p(best), _p(arg), None), lib, "re-score L = 0")
    _refused(lib.dph_rescore(h, 0, _p(x), 4, 2, 10, None, _p(doc), _p(word), _p(first), _p(pred), _p(best), _p(arg), None), lib, "re-score NULL ids")
    # ... and the handle still answers, exactly
    assert lib.dph_search(h, _p(x), 4, 10, _p(D), _p(I)) == 0
    Dr, Ir, D64 = O.flat_ip_search(x, xb, 10, id_base=100)
    ok, msg = O.topk_equivalent(D, I, D64, Ir)
    assert ok, msg
    assert lib.dph_reconstruct(h, 100 + 17, _p(out)) == 0
    np.testing.assert_array_equal(out, O.int8_to_float(xb[17]))
    assert lib.dph_index_destroy(h) == 0


def test_pq_handle_refuses_misuse_and_keeps_working():
    from densephrases_amd._lib import lib
    from oracle import ivfpq_oracle as P
    from tests.test_pq import _index_from_list_numbers, _same_topk, _shard
    rng = np.random.default_rng(4)
    nlist, M, n = 64, 96, 6000
    cent = rng.normal(0, 0.5, (nlist, 768)).astype(np.float32)
    ix, A = _index_from_list_numbers(rng, nlist, M, rng.integers(0, nlist, n), cent)
    h = C.c_void_p()
    _refused(lib.dph_index_create_pq(0, n, 0, M, C.byref(h)), lib, "PQ index with 0 lists")
    _refused(lib.dph_index_create_pq(0, n, nlist, 7, C.byref(h)), lib, "PQ index with M = 7 (768 is not a multiple)")
    _refused(lib.dph_index_create_pq(0, -5, nlist, M, C.byref(h)), lib, "PQ index with a negative code count")
    assert lib.dph_index_create_pq(0, n, nlist, M, C.byref(h)) == 0 and h.value
    x = rng.normal(0, 0.5, (3, 768)).astype(np.float32)
    D = np.zeros((3, 5), np.float32)
    I = np.zeros((3, 5), np.int64)
    _refused(lib.dph_index_finalize(h, None), lib, "finalize before the quantizers / codes were set")
    _refused(lib.dph_search_ivf(h, _p(x), 3, 5, 8, _p(D), _p(I)), lib, "PQ search before finalize")
    sizes = np.full(nlist, n, np.int64)                                     # sums to 64 n, not n
    _refused(lib.dph_index_set_pq_list_sizes(h, _p(sizes)), lib, "list sizes that do not sum to the code count")
    _refused(lib.dph_index_set_pq_list_sizes(h, None), lib, "NULL list sizes")
    assert lib.dph_index_destroy(h) == 0
    # a finished index (through the python loader): bad search arguments, then the right answer
    s = _shard(ix)
    hh = s._h
    _refused(lib.dph_search_ivf(hh, _p(x), 3, 0, 8, _p(D), _p(I)), lib, "PQ search with k = 0")
    _refused(lib.dph_search_ivf(hh, _p(x), 3, 2000, 8, _p(D), _p(I)), lib, "PQ search with k = 2000")
    _refused(lib.dph_search_ivf(hh, None, 3, 5, 8, _p(D), _p(I)), lib, "PQ search with NULL queries")
    _refused(lib.dph_search_prepare_dev(hh, _p(x), 3, 5, None), lib, "the two-stage search on a PQ index")
    out = np.zeros(768, np.float32)
    _refused(lib.dph_reconstruct(hh, 10 ** 12, _p(out)), lib, "PQ reconstruct of an unknown id")
    _refused(lib.dph_index_set_tuning(hh, b"aux", _p(np.asarray([4], np.int32)), 1), lib, "aux rows on a PQ index")
    Dg, Ig = s.search_ivf(x, 5, 8)
    Dr, Ir = P.search(ix, x, 5, 8)
    _same_topk(Dg, Ig, Dr, Ir)
    s.close()
