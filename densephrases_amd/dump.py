"""Phrase-dump / idx2id access for ``MIPS`` (host side, CPU work the reference also does on the CPU).

The reference reads three artefacts (/root/reference/densephrases/index.py:69-122, 246-273):

  <dump_dir>/phrase/*.hdf5            per-document groups: int8 ``start`` rows + ``f2o_start`` +
                                      ``word2char_start/end`` + attrs context/title/offset/scale
  <dump_dir>/start/<index>/idx2id.hdf5   /<offset>/{doc,word} int32 -- the row order of the index
  <dump_dir>/meta_compressed.pkl      blosc-compressed per-document metadata (optional)

``DocStore`` is the in-memory form MIPS consumes: the int8 rows in idx2id order plus per-document metadata.
It can be built from python objects (tests, synthetic dumps), from the ``.npz`` container written by
``DocStore.save_npz`` (the converter output), or from the reference's HDF5 layout through ``densephrases_amd.h5``
(ctypes on libhdf5; python's h5py is not a dependency).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np


@dataclass
class DocMeta:
    doc_idx: int
    title: str
    context: str
    f2o_start: np.ndarray           # int64 [n_f]  filtered -> original token index   (embed_utils.py:130,246)
    word2char_start: np.ndarray     # int32 [n_tok]                                    (embed_utils.py:81-103)
    word2char_end: np.ndarray       # int32 [n_tok]
    start: Optional[np.ndarray] = None   # int8 [n_f, 768]; dropped after upload when not needed


class DocStore:
    def __init__(self, docs: Sequence[DocMeta], rows: Optional[np.ndarray] = None,
                 row2doc: Optional[np.ndarray] = None, row2word: Optional[np.ndarray] = None,
                 offset: float = -2.0, scale: float = 20.0):
        self.docs: Dict[int, DocMeta] = {int(m.doc_idx): m for m in docs}
        self.offset, self.scale = float(offset), float(scale)
        if rows is None:
            # index order = iteration order of the dump's groups: h5py yields string-sorted keys, documents with
            # no vectors are skipped (build_phrase_index.py:196-203)
            ordered = [m for m in sorted(docs, key=lambda m: str(m.doc_idx)) if m.start is not None and m.start.shape[0] > 0]
            rows = np.concatenate([m.start for m in ordered], 0) if ordered else np.zeros((0, 768), np.int8)
            row2doc = np.concatenate([np.full(m.start.shape[0], m.doc_idx, np.int32) for m in ordered]) if ordered else np.zeros(0, np.int32)
            row2word = np.concatenate([np.arange(m.start.shape[0], dtype=np.int32) for m in ordered]) if ordered else np.zeros(0, np.int32)
        self.rows = np.ascontiguousarray(rows, dtype=np.int8)
        self.row2doc = np.ascontiguousarray(row2doc, dtype=np.int32)
        self.row2word = np.ascontiguousarray(row2word, dtype=np.int32)
        assert self.rows.shape[0] == self.row2doc.shape[0] == self.row2word.shape[0]

    @property
    def n_rows(self) -> int:
        return int(self.rows.shape[0])

    def iter_row_blocks(self, lo: int = 0, hi: Optional[int] = None, block: int = 1 << 20,
                        buffers=None) -> Iterator[Tuple[int, np.ndarray]]:
        hi = self.n_rows if hi is None else hi
        for r0 in range(lo, hi, block):
            yield r0, self.rows[r0:min(r0 + block, hi)]

    def doc_starts(self) -> np.ndarray:
        """first stored row of every run of equal doc ids: where a range partition may cut"""
        if self.n_rows == 0:
            return np.zeros(0, np.int64)
        return np.concatenate([[0], np.nonzero(np.diff(self.row2doc))[0] + 1]).astype(np.int64)

    def id_groups(self, lo: int = 0, hi: Optional[int] = None):
        return None                      # in-memory stores number their rows densely

    def f2o_csr(self, lo: int = 0, hi: Optional[int] = None):
        f2o_of = getattr(self.docs, "_f2o", None)            # lazy stores keep f2o_start apart from the heavy metadata
        get = (lambda d: f2o_of[int(d)]) if f2o_of is not None else (lambda d: self.docs[int(d)].f2o_start)
        if lo == 0 and (hi is None or hi >= self.n_rows):
            ids = np.array(sorted(int(d) for d in self.docs.keys()), dtype=np.int32)
        else:                                                # the documents of a row range (range-sharded loading)
            ids = np.array(sorted(set(int(d) for d in self.row2doc[lo:hi].tolist())), dtype=np.int32)
        lens = np.array([len(get(d)) for d in ids], dtype=np.int64)
        off = np.zeros(len(ids) + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        f2o = (np.concatenate([np.asarray(get(d)) for d in ids]).astype(np.int32) if len(ids) else np.zeros(0, np.int32))
        return ids, off, f2o

    def doc_meta(self, doc_idx: int) -> DocMeta:
        if doc_idx not in self.docs:
            raise ValueError("%d not found in dump list" % int(doc_idx))          # index.py:148-156
        return self.docs[doc_idx]

    # ---- converter container
    def save_npz(self, path: str):
        ids = sorted(self.docs.keys())
        np.savez_compressed(
            path, rows=self.rows, row2doc=self.row2doc, row2word=self.row2word,
            codec=np.array([self.offset, self.scale], np.float64), doc_ids=np.array(ids, np.int64),
            titles=np.array([self.docs[d].title for d in ids]), contexts=np.array([self.docs[d].context for d in ids]),
            **{f"f2o_{d}": self.docs[d].f2o_start for d in ids},
            **{f"w2cs_{d}": self.docs[d].word2char_start for d in ids},
            **{f"w2ce_{d}": self.docs[d].word2char_end for d in ids})

    @classmethod
    def load_npz(cls, path: str) -> "DocStore":
        z = np.load(path)
        docs = [DocMeta(int(d), str(z["titles"][i]), str(z["contexts"][i]), z[f"f2o_{d}"], z[f"w2cs_{d}"], z[f"w2ce_{d}"])
                for i, d in enumerate(z["doc_ids"].tolist())]
        return cls(docs, rows=z["rows"], row2doc=z["row2doc"], row2word=z["row2word"],
                   offset=float(z["codec"][0]), scale=float(z["codec"][1]))


def load_dump_and_index(phrase_dump_dir: str, index_path: str, idx2id_path: str):
    """Resolve the reference's three paths to what MIPS loads from.  A ``dump.npz`` converter container next to the
    index wins (a DocStore, whole in host memory); otherwise the reference HDF5 layout is opened lazily through
    libhdf5 (``h5.ReferenceDump``: idx2id in RAM, rows streamed range by range)."""
    index_dir = os.path.dirname(str(index_path))
    for cand in (os.path.join(index_dir, "dump.npz"), str(index_path) if str(index_path).endswith(".npz") else None):
        if cand and os.path.exists(cand):
            return DocStore.load_npz(cand)
    from .h5 import ReferenceDump              # ctypes on libhdf5
    return ReferenceDump(phrase_dump_dir, idx2id_path)
