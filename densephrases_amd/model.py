"""``DensePhrases`` -- the reference's user-facing façade (/root/reference/densephrases/model.py:14-127) over the
MI355X-native ``MIPS``: same constructor arguments, same ``search(query, retrieval_unit, top_k, truecase, return_meta)``
and the same retrieval-unit semantics (model.py:73-97):

    unit        aggregation (MIPS.aggregate_results)   top_k handed to MIPS     what is returned per hit
    phrase      opt1  title_start_end                  top_k                    the answer string
    sentence    opt2  context (+ return_sent)          2 * top_k                the sentence(s) around the answer
    paragraph   opt2  context                          2 * top_k                the paragraph
    document    opt3  title                            2 * top_k                the document title

The query encoder stays PyTorch and stays the caller's: pass ``query2vec`` -- any callable
``list[str] -> list[(start [1,768], end [1,768], tokens)]`` like the reference's ``get_query2vec`` (open_utils.py:83-101),
or one returning a ``[B, 1536]`` torch tensor on the GPU, which then goes to ``MIPS.search_device`` without leaving the
device.  Inside a ``torch.distributed`` job the underlying MIPS is range-sharded over the ranks and ``search`` is a
collective (same result on every rank).
"""
from __future__ import annotations

import logging
import os
from typing import Callable, List, Optional

import numpy as np

from .index import MIPS

logger = logging.getLogger(__name__)

_AGG = {"phrase": "opt1", "sentence": "opt2", "paragraph": "opt2", "document": "opt3"}        # model.py:73


class DensePhrases(object):
    def __init__(self, load_dir=None, dump_dir=None, index_name="start/1048576_flat_OPQ96", device="cuda", verbose=False,
                 query2vec: Optional[Callable] = None, truecase=None, phrase_dir="phrase", index_path="index.faiss",
                 idx2id_path="idx2id.hdf5", mips: Optional[MIPS] = None, **kwargs):
        """``load_dir`` (the query-encoder checkpoint) is accepted for signature compatibility; this package does not
        re-implement the SpanBERT encoder (it "stays PyTorch-ROCm"): hand the encoder in as ``query2vec``.  The index
        paths are resolved like ``load_phrase_index`` (open_utils.py:26-43)."""
        if not verbose:
            logging.getLogger("densephrases_amd").setLevel(logging.WARNING)
        self.load_dir, self.dump_dir, self.index_name = load_dir, dump_dir, index_name
        if mips is None:
            index_dir = os.path.join(dump_dir, index_name)
            mips = MIPS(phrase_dump_dir=os.path.join(dump_dir, phrase_dir), index_path=os.path.join(index_dir, index_path),
                        idx2id_path=os.path.join(index_dir, idx2id_path), cuda=(device == "cuda"),
                        logging_level=logging.INFO if verbose else logging.WARNING)
        self.mips = mips
        self.query2vec = query2vec
        self.truecase = truecase

    @classmethod
    def from_parts(cls, mips: MIPS, query2vec: Callable, truecase=None):
        return cls(mips=mips, query2vec=query2vec, truecase=truecase)

    def set_encoder(self, query2vec: Callable):
        """model.py:111-116 loads a checkpoint; here the caller supplies the callable"""
        self.query2vec = query2vec

    def search(self, query="", retrieval_unit="phrase", top_k=10, truecase=True, return_meta=False):
        single_query = False
        if type(query) == str:                                                   # model.py:57-63
            batch_query, single_query = [query], True
        else:
            assert type(query) == list
            batch_query = query
        if truecase and self.truecase is not None:                               # model.py:66-68
            batch_query = [self.truecase.get_true_case(q) if q == q.lower() else q for q in batch_query]
        if self.query2vec is None:
            raise RuntimeError("DensePhrases.search: no query encoder (pass query2vec= or call set_encoder)")
        outs = self.query2vec(batch_query)
        if retrieval_unit not in _AGG:                                           # model.py:75-76
            raise NotImplementedError(f'"{retrieval_unit}" not supported. Choose one of {_AGG.keys()}.')
        search_top_k = top_k * 2 if retrieval_unit in ("sentence", "paragraph", "document") else top_k   # model.py:77-79
        kw = dict(q_texts=batch_query, top_k=search_top_k, max_answer_length=10, aggregate=True,
                  agg_strat=_AGG[retrieval_unit], return_sent=retrieval_unit == "sentence")
        if _is_device_tensor(outs):
            rets = self.mips.search_device(outs, **kw)                           # the query never leaves the GPU
        else:
            start = np.concatenate([np.asarray(o[0]) for o in outs], 0)          # model.py:70-72
            end = np.concatenate([np.asarray(o[1]) for o in outs], 0)
            rets = self.mips.search(np.concatenate([start, end], 1), nprobe=256, return_idxs=False, **kw)
        rets = [ret[:top_k] for ret in rets]                                     # model.py:88-98
        if retrieval_unit == "phrase":
            retrieved = [[rr["answer"] for rr in ret][:top_k] for ret in rets]
        elif retrieval_unit in ("sentence", "paragraph"):
            retrieved = [[rr["context"] for rr in ret][:top_k] for ret in rets]
        else:
            retrieved = [[rr["title"][0] for rr in ret][:top_k] for ret in rets]
        if single_query:
            rets, retrieved = rets[0], retrieved[0]
        return (retrieved, rets) if return_meta else retrieved


def _is_device_tensor(x) -> bool:
    try:
        import torch
        return isinstance(x, torch.Tensor) and x.is_cuda
    except Exception:
        return False
