"""A ``faiss`` module for the reference's UNMODIFIED ``densephrases/index.py`` (level 2): exactly the
slice of the FAISS python API that file touches, backed by a libdph shard resident in HBM.

    import densephrases_amd.faiss_compat as fc;  fc.install()      # sys.modules['faiss'] = this module
    from densephrases.index import MIPS                            # the reference's class, as written

What index.py uses (file:line under /root/reference/densephrases/) and what answers here:

    faiss.read_index(index_path, faiss.IO_FLAG_ONDISK_SAME_DIR)          :30   -> DphIndex: a real FAISS file (OPQ + IVFPQ, parsed by
                                                                                  faiss_io.py) becomes a PQ index in HBM; otherwise the
                                                                                  rows of <dump_dir>/phrase in idx2id order
    faiss.downcast_index(self.index.index).reconstruct                    :31   -> DphIndex.reconstruct
    faiss.vector_to_array(faiss.downcast_VectorTransform(
        self.index.chain.at(0)).A).reshape(d, d)                          :32   -> the OPQ matrix of the file / identity over the raw dump
    self.index.ntotal / self.index.d                                      :32,34,128
    faiss.extract_index_ivf(self.index) ; .nprobe = 256 ; .quantizer      :52-62 -> nprobe of the PQ index (accepted, without effect, over
                                                                                  the raw dump: that search is exact)
    faiss.index_cpu_to_all_gpus(quantizer)                                :55   -> returned as is
    self.index.search(query_concat, top_k)                                :200  -> dph_search (IVFPQ ADC / exact IP, FAISS padding)
    self.reconst_fn(id)  (raises on unknown ids)                          :286,296 -> dph_reconstruct

``index_path`` is ``<dump_dir>/<index_name>/index.faiss`` (open_utils.py:26-33).  When that file is not a FAISS index (empty,
absent, or a flat index of the de-quantised dump) the index *is* the int8 dump: rows come from ``<dump_dir>/phrase/*.hdf5`` in
the order of ``idx2id.hdf5`` next to the index.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

from . import _lib

IO_FLAG_ONDISK_SAME_DIR = 0x8
IO_FLAG_MMAP = 0x2


class _Transform:
    def __init__(self, d, A=None):
        self.A = (np.eye(d, dtype=np.float32) if A is None else np.asarray(A, np.float32)).reshape(-1)
        self.d_in = self.d_out = d


class _Chain:
    def __init__(self, d, A=None):
        self._t = _Transform(d, A)

    def at(self, i):
        if i != 0:
            raise IndexError("one transform")
        return self._t

    def size(self):
        return 1


class DphIndex:
    """quacks like the ``faiss.IndexPreTransform`` (over an IVF index) index.py holds"""

    def __init__(self, shard: "_lib.Shard", store=None):
        self._shard, self._store = shard, store
        self.d = shard.d
        self.ntotal = shard.ntotal
        self.index = self                # faiss.downcast_index(self.index.index)            (index.py:31)
        self._is_pq = getattr(shard, "pq", None) is not None
        # self.index.chain.at(0).A (index.py:32): identity over the raw dump, the OPQ matrix of a real IVFPQ index
        self.chain = _Chain(self.d, shard.transform() if self._is_pq else None)
        self._nprobe = 256               # index_ivf.nprobe = 256 (index.py:53,62): the nprobe of a PQ index, accepted and
                                         # without effect over the raw dump (that search is exact)
        self.quantizer = types.SimpleNamespace(ntotal=0)     # index_ivf.quantizer (index.py:54-56)
        self.is_trained = True

    @property
    def nprobe(self):
        return self._nprobe

    @nprobe.setter
    def nprobe(self, v):
        self._nprobe = int(v)
        if self._is_pq:
            self._shard.set_tuning("nprobe", max(1, int(v)))

    def search(self, x, k):              # index.py:200
        try:
            return self._shard.search(np.asarray(x, dtype=np.float32), int(k))
        except _lib.DphError as e:       # FAISS raises RuntimeError through SWIG
            raise RuntimeError(str(e))

    def reconstruct(self, i):            # index.py:286,296 -- FAISS throws on unknown ids, the caller substitutes zeros
        try:
            return self._shard.reconstruct(int(i))
        except _lib.DphError as e:
            raise RuntimeError(str(e))


def index_from_store(store, device: int = 0) -> DphIndex:
    """a resident index from anything with rows / idx2id / f2o (a DocStore or an h5.ReferenceDump)"""
    shard = _lib.Shard(store.n_rows, device=device)
    shard.set_codec(store.offset, store.scale)
    for r0, rows in store.iter_row_blocks():
        shard.upload(rows, r0)
    shard.set_idx2id(store.row2doc, store.row2word)
    shard.set_f2o(*store.f2o_csr())
    groups = store.id_groups() if hasattr(store, "id_groups") else None
    if groups is not None:
        shard.set_id_groups(*groups)
    shard.finalize()
    return DphIndex(shard, store)


def pq_index_from_file(parsed, idx2id_path, phrase_dir, device: int = 0) -> DphIndex:
    """a real IndexPreTransform + IndexIVFPQ file (densephrases_amd.faiss_io) resident in HBM"""
    from .dump import load_dump_and_index
    store = load_dump_and_index(phrase_dir, "", idx2id_path)
    shard = _lib.Shard.from_faiss_index(parsed, device=device)
    if store.n_rows != shard.ntotal:
        raise RuntimeError(f"index.faiss holds {shard.ntotal} vectors, idx2id {store.n_rows}")
    shard.set_idx2id(store.row2doc, store.row2word)
    groups = store.id_groups() if hasattr(store, "id_groups") else None
    if groups is not None:
        shard.set_id_groups(*groups)
    shard.set_f2o(*store.f2o_csr())
    shard.finalize()
    return DphIndex(shard, store)


def read_index(path, io_flags=0):
    index_dir = os.path.dirname(str(path))
    from . import faiss_io
    if faiss_io.looks_like_faiss_index(str(path)):
        parsed = faiss_io.read_index(str(path), io_flags | IO_FLAG_ONDISK_SAME_DIR)
        if not isinstance(parsed, faiss_io.FlatIndex):
            dump_dir = os.path.dirname(os.path.dirname(index_dir))
            return pq_index_from_file(parsed, os.path.join(index_dir, "idx2id.hdf5"), os.path.join(dump_dir, "phrase"),
                                      device=int(os.environ.get("DPH_DEVICE", "0")))
    dump_dir = os.path.dirname(os.path.dirname(index_dir))           # <dump_dir>/start/<name>/index.faiss
    from .dump import load_dump_and_index
    store = load_dump_and_index(os.path.join(dump_dir, "phrase"), str(path), os.path.join(index_dir, "idx2id.hdf5"))
    return index_from_store(store, device=int(os.environ.get("DPH_DEVICE", "0")))


def downcast_index(index):
    return index


def downcast_VectorTransform(vt):
    return vt


def vector_to_array(v):
    return np.asarray(v)


def extract_index_ivf(index):
    return index


def index_cpu_to_all_gpus(index, co=None, ngpu=-1):
    return index


def install():
    """make ``import faiss`` resolve to this module (eval_phrase_retrieval.py:12, train_query.py:12, index.py:1)"""
    sys.modules["faiss"] = sys.modules[__name__]
    return sys.modules[__name__]
