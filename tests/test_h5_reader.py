"""The ctypes/libhdf5 reader against files written by h5py in the reference's layout (CPU only).  Needs an
interpreter with h5py to WRITE the fixture (/opt/conda/bin/python3.9 in this image); skipped where absent."""
import os
import pickle
import subprocess

import numpy as np
import pytest

from tests._golden import GOLD, load_toy_docs

PY39 = "/opt/conda/bin/python3.9"


@pytest.fixture(scope="module")
def ref_layout(tmp_path_factory):
    if not os.path.exists(PY39):
        pytest.skip("no interpreter with h5py to write the fixture")
    out = tmp_path_factory.mktemp("dump")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([PY39, os.path.join(here, "_make_h5_dump.py"), os.path.join(GOLD, "toy_dump.npz"), str(out)],
                       capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("h5py writer failed: " + r.stderr[-300:])
    return str(out)


def test_reference_layout_roundtrip(ref_layout):
    from densephrases_amd.h5 import load_reference_layout
    from oracle.mips_oracle import build_index_from_docs
    docs = load_toy_docs()
    want = build_index_from_docs(docs)
    store = load_reference_layout(os.path.join(ref_layout, "phrase"),
                                  os.path.join(ref_layout, "start", "toy_flat_none", "idx2id.hdf5"))
    np.testing.assert_array_equal(store.rows, want.xb)
    np.testing.assert_array_equal(store.row2doc, want.row2doc)
    np.testing.assert_array_equal(store.row2word, want.row2word)
    assert store.offset == -2.0 and store.scale == 20.0
    for m in docs:
        g = store.doc_meta(m.doc_idx)
        assert g.title == m.title and g.context == m.context
        np.testing.assert_array_equal(g.f2o_start, m.f2o_start)
        np.testing.assert_array_equal(g.word2char_start, m.word2char_start)
        np.testing.assert_array_equal(g.word2char_end, m.word2char_end)
    ids, off, f2o = store.f2o_csr()
    assert list(ids) == sorted(m.doc_idx for m in docs) and off[-1] == f2o.shape[0]
    with pytest.raises(ValueError):
        store.doc_meta(123456)


def test_meta_compressed_pkl_via_libblosc(ref_layout):
    """meta_compressed.pkl (compress_metadata.py:45-53) is preferred over the HDF5 groups when present."""
    import ctypes as C
    from densephrases_amd import h5
    path = h5._find("blosc")
    if not path:
        pytest.skip("libblosc not present")
    B = C.CDLL(path)
    B.blosc_compress.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_char_p, C.c_void_p, C.c_size_t]

    def comp(b: bytes) -> bytes:
        out = C.create_string_buffer(len(b) + 64)
        n = B.blosc_compress(5, 1, 1, len(b), b, out, len(b) + 64)
        assert n > 0
        return out.raw[:n]

    assert h5.blosc_decompress(comp(b"hello world" * 10)) == b"hello world" * 10
    docs = load_toy_docs()
    meta = {str(m.doc_idx): {"word2char_start": comp(m.word2char_start.tobytes()), "word2char_end": comp(m.word2char_end.tobytes()),
                             "f2o_start": comp(m.f2o_start.tobytes()), "context": comp(("META:" + m.context).encode()),
                             "title": m.title, "dtypes": {"word2char_start": m.word2char_start.dtype,
                                                          "word2char_end": m.word2char_end.dtype, "f2o_start": m.f2o_start.dtype}}
            for m in docs}
    with open(os.path.join(ref_layout, "meta_compressed.pkl"), "wb") as f:
        pickle.dump(meta, f)
    try:
        store = h5.load_reference_layout(os.path.join(ref_layout, "phrase"),
                                         os.path.join(ref_layout, "start", "toy_flat_none", "idx2id.hdf5"))
        g = store.doc_meta(docs[0].doc_idx)
        assert g.context == "META:" + docs[0].context
        np.testing.assert_array_equal(g.f2o_start, docs[0].f2o_start)
    finally:
        os.remove(os.path.join(ref_layout, "meta_compressed.pkl"))


def test_reference_dump_streams_ranges_and_merged_index_groups(tmp_path):
    """ReferenceDump: rows streamed in document-aligned blocks (into caller buffers), f2o CSR of a row range, and the
    idx2id of a MERGED index (two sub-indexes, id offsets 0 and 10^8: index.py:135-140) concatenated into dense rows
    with the id <-> row translation MIPS hands to libdph."""
    if not os.path.exists(PY39):
        pytest.skip("no interpreter with h5py to write the fixture")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([PY39, os.path.join(here, "_make_h5_dump.py"), os.path.join(GOLD, "toy_dump.npz"), str(tmp_path),
                        "split:100000000"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("h5py writer failed: " + r.stderr[-300:])
    from densephrases_amd.h5 import ReferenceDump
    from densephrases_amd.dist import partition_rows
    from oracle.mips_oracle import build_index_from_docs
    want = build_index_from_docs(load_toy_docs())
    dump = ReferenceDump(os.path.join(str(tmp_path), "phrase"), os.path.join(str(tmp_path), "start", "toy_flat_none", "idx2id.hdf5"))
    n = dump.n_rows
    assert n == want.xb.shape[0] and not dump.single_dense_group
    np.testing.assert_array_equal(dump.row2doc, want.row2doc)
    assert list(dump.id_offsets) == [0, 100000000] and dump.row_starts[0] == 0 and dump.row_starts[-1] == n
    cut = int(dump.row_starts[1])
    # ids <-> rows
    rows = np.array([0, cut - 1, cut, n - 1])
    ids = dump.ids_of_rows(rows)
    np.testing.assert_array_equal(ids, [0, cut - 1, 100000000, 100000000 + n - 1 - cut])
    np.testing.assert_array_equal(dump.rows_of_ids(ids), rows)
    np.testing.assert_array_equal(dump.rows_of_ids(np.array([cut, 99999999, 100000000 + n - cut, -5])), [-1, -1, -1, -1])
    # streamed blocks: whole documents, small blocks, caller-owned buffers reused in turn
    bufs = [np.empty((64, 768), np.int8) for _ in range(2)]
    got = np.empty_like(want.xb)
    seen = 0
    for r0, blk in dump.iter_row_blocks(0, n, block=64, buffers=bufs):
        assert r0 in set(dump.doc_starts().tolist())
        got[r0:r0 + blk.shape[0]] = blk
        seen += blk.shape[0]
    assert seen == n
    np.testing.assert_array_equal(got, want.xb)
    # a 2-way range partition at document boundaries: per-shard rows, CSR and id groups
    parts = partition_rows(n, 2, doc_starts=dump.doc_starts())
    assert parts[0][1] == parts[1][0] and parts[0][1] in set(dump.doc_starts().tolist())
    for lo, hi in parts:
        ids_, off, f2o = dump.f2o_csr(lo, hi)
        assert set(ids_.tolist()) == set(dump.row2doc[lo:hi].tolist()) and off[-1] == f2o.shape[0]
        go, gs = dump.id_groups(lo, hi)
        assert gs[0] == 0 and gs[-1] == hi - lo
        local = np.arange(hi - lo)
        g = np.searchsorted(gs, local, side="right") - 1
        np.testing.assert_array_equal(go[g] + local - gs[g], dump.ids_of_rows(local + lo))
    with pytest.raises(ValueError):
        list(dump.iter_row_blocks(1, n))            # not a document boundary


def test_packed_row_cache_replaces_the_hdf5_stream_on_the_next_load(ref_layout, tmp_path):
    """MIPS(cache_dir=...): the first load of a row range records what the HDF5 reads deliver (rows + f2o CSR), later loads
    of the same range over unchanged artefacts never touch a document of the dump; a changed artefact, another range or a
    partial recording do not count."""
    from densephrases_amd.h5 import ReferenceDump
    phrase, idx = os.path.join(ref_layout, "phrase"), os.path.join(ref_layout, "start", "toy_flat_none", "idx2id.hdf5")
    cache = str(tmp_path / "packed")

    def load(dump, lo, hi, block=64):
        rows = np.zeros((hi - lo, 768), np.int8)
        for r0, blk in dump.iter_row_blocks(lo, hi, block=block):
            rows[r0 - lo:r0 - lo + blk.shape[0]] = blk
        return rows, dump.f2o_csr(lo, hi)

    d0 = ReferenceDump(phrase, idx)
    n = d0.n_rows
    cut = int(d0.doc_starts()[len(d0.doc_starts()) // 2])
    want_rows, want_f2o = load(d0, 0, n)

    d1 = ReferenceDump(phrase, idx)
    assert d1.attach_row_cache(cache, 0, n) is False              # nothing there yet: record
    got_rows, got_f2o = load(d1, 0, n)
    np.testing.assert_array_equal(got_rows, want_rows)
    assert d1.finish_row_cache() is True
    assert sorted(os.listdir(cache)) == [f"rows_0_{n}.f2o.npz", f"rows_0_{n}.i8", f"rows_0_{n}.json"]

    d2 = ReferenceDump(phrase, idx)
    assert d2.attach_row_cache(cache, 0, n) is True
    d2._read_rows_hdf5 = None                                      # any HDF5 row read would now raise
    d2.f2o_of = None
    got_rows, got_f2o = load(d2, 0, n, block=100)
    np.testing.assert_array_equal(got_rows, want_rows)
    for a, b in zip(got_f2o, want_f2o):
        np.testing.assert_array_equal(a, b)
    assert d2.finish_row_cache() is True

    # document metadata still comes from the dump, on demand
    d3 = ReferenceDump(phrase, idx)
    assert d3.attach_row_cache(cache, 0, n) is True
    m = d3.doc_meta(int(d3.row2doc[0]))
    np.testing.assert_array_equal(m.f2o_start, d0.doc_meta(int(d0.row2doc[0])).f2o_start)

    # another range: its own copy; a partial recording is dropped
    d4 = ReferenceDump(phrase, idx)
    assert d4.attach_row_cache(cache, cut, n) is False
    next(iter(d4.iter_row_blocks(cut, n, block=32)))
    d4.f2o_csr(cut, n)
    assert d4.finish_row_cache() is False
    assert not os.path.exists(os.path.join(cache, f"rows_{cut}_{n}.i8"))
    assert not [f for f in os.listdir(cache) if "tmp" in f]
    # ... but its f2o table was complete and is kept: what a PQ index (rows = False) asks for
    d5 = ReferenceDump(phrase, idx)
    assert d5.attach_row_cache(cache, cut, n, rows=False) is True
    d5.f2o_of = None
    for a, b in zip(d5.f2o_csr(cut, n), d0.f2o_csr(cut, n)):
        np.testing.assert_array_equal(a, b)
    assert d5.finish_row_cache() is True
    # an f2o-only attach leaves the rows of the whole-range copy in place
    d6 = ReferenceDump(phrase, idx)
    assert d6.attach_row_cache(cache, 0, n, rows=False) is True and d6.finish_row_cache() is True
    d6 = ReferenceDump(phrase, idx)
    assert d6.attach_row_cache(cache, 0, n) is True

    # a touched artefact invalidates the copy (the fingerprint holds sizes and mtimes)
    os.utime(idx, ns=(1, 1))
    d7 = ReferenceDump(phrase, idx)
    assert d7.attach_row_cache(cache, 0, n, write=False) is False
    got_rows, _ = load(d7, 0, n)                                   # falls back to the HDF5 stream
    np.testing.assert_array_equal(got_rows, want_rows)
    with pytest.raises(ValueError):
        d7.attach_row_cache(cache, 0, cut + 1)                     # ranges are cut at document boundaries


def test_row_cache_recordings_of_other_hosts_are_left_alone(ref_layout, tmp_path):
    """A cache_dir on a shared file system: recordings carry the writer's HOST and pid.  A fresh recording of another
    host -- whose pid means nothing here -- is not reaped by this host's start (only after a day without a write); this host's dead
    pids and stale foreign / old-format files are; and a writer whose recording was removed under it still finishes its load (the copy
    is simply not published)."""
    import socket
    import time
    from densephrases_amd.h5 import ReferenceDump
    phrase, idx = os.path.join(ref_layout, "phrase"), os.path.join(ref_layout, "start", "toy_flat_none", "idx2id.hdf5")
    cache = str(tmp_path / "packed")
    os.makedirs(cache)
    d0 = ReferenceDump(phrase, idx)
    n = d0.n_rows
    base = os.path.join(cache, f"rows_0_{n}.i8.tmp")
    host = "".join(ch if ch.isalnum() or ch in "-_" else "_" for ch in socket.gethostname())[:48] or "host"
    foreign_live, foreign_stale, own_dead, old_format = base + "otherhost.4242", base + "otherhost.77", base + f"{host}.999999999", base + "31337"
    for p in (foreign_live, foreign_stale, own_dead, old_format):
        open(p, "wb").write(b"x")
    two_days_ago = time.time() - 2 * 86400
    os.utime(foreign_stale, (two_days_ago, two_days_ago))
    os.utime(old_format, (two_days_ago, two_days_ago))
    d1 = ReferenceDump(phrase, idx)
    assert d1.attach_row_cache(cache, 0, n) is False
    assert os.path.exists(foreign_live) and not os.path.exists(foreign_stale) and not os.path.exists(own_dead) and not os.path.exists(old_format)
    mine = [f for f in os.listdir(cache) if f.startswith(f"rows_0_{n}.i8.tmp{host}.")]
    assert mine == [f"rows_0_{n}.i8.tmp{host}.{os.getpid()}"]
    for _ in d1.iter_row_blocks(0, n, block=64):
        pass
    d1.f2o_csr(0, n)
    os.unlink(os.path.join(cache, mine[0]))                         # somebody (another host's clean-up of round 5) removed it
    assert d1.finish_row_cache() is False                           # no exception: the rows were not published, the f2o table was
    assert os.path.exists(os.path.join(cache, f"rows_0_{n}.f2o.npz")) and not os.path.exists(os.path.join(cache, f"rows_0_{n}.i8"))
