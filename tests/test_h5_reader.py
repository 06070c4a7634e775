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
