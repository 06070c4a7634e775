"""``MIPS`` -- drop-in for the reference's ``densephrases.index.MIPS`` (/root/reference/densephrases/index.py:23-482)
with the FAISS search, the per-candidate ``reconstruct`` loop and the torch window re-scoring replaced by libdph's
HIP kernels over an int8 phrase shard resident in HBM.

Same constructor, same ``search`` signature, same result dictionaries (``context, title, doc_idx, start_pos,
end_pos, start_idx, end_idx, score, start_vec, end_vec, answer``), same aggregation strategies.  What stays in
python is what the reference also does in python and is string work: metadata lookup, dict assembly, paragraph /
sentence cropping, de-duplication.

There is no CPU path: the module imports ``densephrases_amd._lib`` which raises if libdph.so is not built.
"""
from __future__ import annotations

import logging
import os
import re
import string
from time import time
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from . import _dph_host          # C++ host half (csrc/dph_host.cpp), built in-tree by densephrases_amd/build.py; no fallback
from .dump import DocStore, load_dump_and_index

logging.basicConfig(format="%(asctime)s - %(levelname)s - %(name)s -   %(message)s", datefmt="%m/%d/%Y %H:%M:%S",
                    level=logging.INFO)
logger = logging.getLogger(__name__)

_DUMMY_SCORE = -1e8       # index.py:400-401, 441
_DROP_BELOW = -1e5        # index.py:420, 447


def normalize_answer(s: str) -> str:
    """DrQA-style answer normalisation used by agg_strat 'opt4' (eval_utils.py:9-24)."""
    s = "".join(ch for ch in s.lower() if ch not in set(string.punctuation))
    s = re.sub(r"\b(a|an|the)\b", " ", s)
    return " ".join(s.split())


from .sentencizer import split_sentences  # noqa: E402,F401  (the sentence units of return_sent: index.py:65-66,178-187)


class _IndexView:
    """What callers read off ``mips.index`` (``.ntotal``, ``.d``; eval_phrase_retrieval.py, index.py:34)."""

    def __init__(self, shard: _lib.Shard, ntotal: Optional[int] = None):
        self._s = shard
        self._ntotal = ntotal

    @property
    def ntotal(self):
        """rows of the WHOLE index (all ranks), like faiss Index.ntotal"""
        return self._s.ntotal if self._ntotal is None else self._ntotal

    @property
    def d(self):
        return self._s.d

    def search(self, x, k):
        return self._s.search(x, k)

    def reconstruct(self, i):
        return self._s.reconstruct(i)


def _default_dist():
    """(rank, world, dist) of the running torch.distributed job, or (0, 1, None)."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size(), dist
    except Exception:
        pass
    return 0, 1, None


class MIPS(object):
    def __init__(self, phrase_dump_dir, index_path, idx2id_path, cuda=False, logging_level=logging.INFO,
                 device: Optional[int] = None, _store=None, rank: Optional[int] = None, world: Optional[int] = None,
                 dist=None, ivf: Optional[dict] = None, cache_dir: Optional[str] = None, force_collectives: bool = False):
        """Same arguments as the reference (index.py:24).  ``cuda`` is accepted for compatibility; the search always
        runs on the GPU.  ``index_path`` names the reference's ``index.faiss``; no FAISS file is read -- the index
        *is* the int8 dump in idx2id row order (build_phrase_index.py:192-276).

        Range-sharded over the GPUs of a node: inside a ``torch.distributed`` job (or with explicit
        ``rank`` / ``world`` / ``dist``) every rank constructs MIPS with the same arguments, loads ONLY its
        document-aligned row range (streamed: phrase/*.hdf5 -> two pinned staging buffers -> HBM, never whole in host
        memory) and ``search`` becomes a collective call that returns the same merged result on every rank.

        ``ivf={"nlist": 4096, "nprobe": 256[, "centroids": float32 [nlist,768], "iters": 10, "train_rows": None]}``
        stores the shard LIST-MAJOR behind a coarse quantizer (the IVF half of the reference's IndexIVFPQ with exact
        in-list scores: build_phrase_index.py:96-153, index.py:52-62): k-means + list assignment on the GPU
        (densephrases_amd/ivf.py), then every search -- ``nprobe`` of ``search`` / ``search_dense`` included -- scores
        the rows of the probed lists only.  The permutation happens in HBM (the rows are there twice while it runs);
        ranks of a multi-GPU job must share ``centroids``.

        ``cache_dir`` (default: the environment's ``DPH_DUMP_CACHE``, unset = off): the first load of a row range leaves a
        packed copy of its rows and f2o table there (h5.ReferenceDump.attach_row_cache); later starts of the same range
        over unchanged artefacts stream that copy instead of touching every document of the HDF5 dump."""
        logger.setLevel(logging_level)
        self.phrase_dump_dir = phrase_dump_dir
        self.index_path = index_path
        self.max_idx = int(1e8) if "PQ" not in str(index_path) else int(1e9)     # index.py:33
        self.cuda = True
        self.num_docs_list: List[float] = []
        t0 = time()
        d_rank, d_world, d_dist = _default_dist()
        self.rank = d_rank if rank is None else int(rank)
        self.world = d_world if world is None else int(world)
        self.dist = d_dist if dist is None else dist
        # a job of ONE rank normally skips every collective; `force_collectives` sends the exchanges through `dist` all the same
        # (tests/test_dist_nccl.py: RCCL communicator, dtypes and stream ordering on a 1-GPU box)
        self.force_collectives = bool(force_collectives) and self.dist is not None
        if self.world > 1 and self.dist is None:
            raise ValueError("MIPS: world > 1 needs a torch.distributed process group (or a `dist` object)")
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0")) if self.world > 1 else 0
        store = _store if _store is not None else load_dump_and_index(phrase_dump_dir, index_path, idx2id_path)
        self.store = store
        self.ivf = None
        self._cache_dir = cache_dir if cache_dir is not None else (os.environ.get("DPH_DUMP_CACHE") or None)
        if not hasattr(store, "attach_row_cache"):
            self._cache_dir = None
        from . import faiss_io
        if _store is None and faiss_io.looks_like_faiss_index(str(index_path)):
            parsed = faiss_io.read_index(str(index_path), faiss_io.IO_FLAG_ONDISK_SAME_DIR)        # index.py:30
            if not isinstance(parsed, faiss_io.FlatIndex):
                # the reference's own index type: OPQ + IVFPQ codes resident in HBM instead of the raw int8 rows
                self._init_pq(parsed, store, device, ivf)
                logger.info(f"index ntotal: {self.index.ntotal} | PQ codes resident on GPU {device} | load {time() - t0:.1f}s")
                return
            # fine_quant 'none' (build_phrase_index.py:117-118): a flat fp32 index of the de-quantised dump rows -- the
            # resident int8 dump below IS that index
        n = store.n_rows
        from .dist import partition_rows
        self.row_lo, self.row_hi = partition_rows(n, self.world, doc_starts=store.doc_starts())[self.rank]
        lo, hi = self.row_lo, self.row_hi
        groups = store.id_groups(lo, hi)
        cached = self._cache_dir is not None and store.attach_row_cache(self._cache_dir, lo, hi)
        try:
            if ivf is not None:
                self._build_ivf(store, lo, hi, groups, device, dict(ivf))
            else:
                self.shard = _lib.Shard(hi - lo, device=device, id_base=lo)
                self.shard.set_codec(store.offset, store.scale)
                self._upload(store, lo, hi)
            self.shard.set_idx2id(store.row2doc[lo:hi], store.row2word[lo:hi])
            self.shard.set_f2o(*store.f2o_csr(lo, hi))
        except BaseException:
            # a failed load must not leave its (pid-named, up to shard-sized) recording in the cache directory
            if self._cache_dir is not None:
                store.abort_row_cache()
            raise
        if self._cache_dir is not None:
            kept = store.finish_row_cache()
            logger.info(f"packed copy of rows [{lo}, {hi}) under {self._cache_dir}: {'read' if cached else 'written' if kept else 'not written'}")
        if groups is not None and self.ivf is None:
            self.shard.set_id_groups(*groups)
        self.shard.finalize()
        if self.ivf is not None:
            self.shard.set_tuning("nprobe", self.ivf["nprobe"])
        self.index = _IndexView(self.shard, n)
        self.R = np.eye(self.shard.d, dtype=np.float32)      # flat index: no OPQ rotation (index.py:32)
        logger.info(f"index ntotal: {self.index.ntotal} | rows [{lo}, {hi}) resident on GPU {device} "
                    f"(rank {self.rank}/{self.world}) | load {time() - t0:.1f}s")

    def _init_pq(self, parsed, store, device: int, ivf):
        """``index_path`` is a real FAISS file holding IndexPreTransform(OPQMatrix) -> IndexIVFPQ (build_phrase_index.py:
        108-116, what ``1048576_flat_OPQ96`` names): codes, codebooks, coarse centroids and the OPQ matrix go to HBM
        (csrc/dph_pq.hip); search = FAISS' IVFPQ search, windows over reconstructed vectors un-rotated by R (index.py:
        282-302, 340, 365).  The dump's metadata (idx2id, f2o, documents) is used as for a raw-dump shard; its int8 rows
        are not read.  Range-sharded (world > 1) like the raw dump: north_star "(or PQ-compressed) phrase dump ... range-partitioned"."""
        if ivf is not None:
            raise ValueError("MIPS(ivf=...) builds lists over the raw dump; a FAISS index file brings its own")
        n = store.n_rows
        if int(parsed.ntotal) != n:
            raise ValueError(f"MIPS: index.faiss holds {parsed.ntotal} vectors, idx2id {n}")
        # world > 1: every rank holds the OPQ matrix, all coarse centroids and the codebooks, and the CODES of its own row range (cut
        # at document boundaries like the raw dump, so a candidate's window stays on its rank): every rank probes the same lists,
        # scores its share of them, and the ranks' top-k merge to the single-GPU answer (dist.ShardedSearcher without the union bound)
        from .dist import partition_rows
        self.row_lo, self.row_hi = partition_rows(n, self.world, doc_starts=store.doc_starts())[self.rank]
        lo, hi = self.row_lo, self.row_hi
        keep = None
        if self.world > 1:
            rows_of = getattr(store, "rows_of_ids", None)
            def keep(ids, lo=lo, hi=hi, rows_of=rows_of):
                rows = rows_of(ids) if rows_of is not None else ids
                return (rows >= lo) & (rows < hi)
        self.shard = _lib.Shard.from_faiss_index(parsed, device=device, keep=keep)
        if self.shard.n_rows != hi - lo:
            raise ValueError(f"MIPS: the index holds {self.shard.n_rows} codes for the rows [{lo}, {hi}) of this rank")
        self.shard.set_idx2id(store.row2doc[lo:hi], store.row2word[lo:hi])
        groups = store.id_groups(lo, hi)
        if groups is None and self.world > 1:
            groups = (np.asarray([lo], np.int64), np.asarray([0, hi - lo], np.int64))       # ids = lo + local row
        if groups is not None:
            self.shard.set_id_groups(*groups)
        if self._cache_dir is not None:
            store.attach_row_cache(self._cache_dir, lo, hi, rows=False)
        self.shard.set_f2o(*store.f2o_csr(lo, hi))
        if self._cache_dir is not None:
            store.finish_row_cache()
        self.shard.finalize()
        self.pq = dict(self.shard.pq)
        if self.pq["nprobe"] > self.pq["nlist"]:               # an index with fewer lists than the reference's nprobe = 256 (index.py:53)
            self.pq["nprobe"] = self.pq["nlist"]
            self.shard.set_tuning("nprobe", self.pq["nprobe"])
        self.index = _IndexView(self.shard, n)
        self.R = self.shard.transform()                       # index.py:32

    def _build_ivf(self, store, lo: int, hi: int, groups, device: int, ivf: dict):
        """List-major shard of rows [lo, hi), built where the rows lie: they are streamed into HBM like a flat shard,
        then -- all on the GPU (densephrases_amd/ivf.py: make_list_major_resident) -- centroids (given, or Lloyd
        iterations over a sample of this range), list of a row = arg-max inner product (fused MFMA GEMM + arg-max over
        the resident int8 rows, near-ties re-ranked in float64), rows sorted by (list, id) and gathered into contiguous
        lists padded to whole tiles."""
        import torch
        from .ivf import make_list_major_resident
        if groups is not None:
            raise ValueError("MIPS(ivf=...): a merged multi-offset index cannot be stored list-major (ids are not contiguous)")
        nlist = int(ivf["nlist"])
        cent = ivf.get("centroids")
        if cent is None and self.world > 1:
            raise ValueError("MIPS(ivf=...): the ranks of a multi-GPU job must be given the same `centroids`")
        if cent is not None and tuple(np.shape(cent)) != (nlist, _lib.DIM):
            raise ValueError(f"MIPS(ivf=...): centroids must be [{nlist}, {_lib.DIM}]")
        torch.cuda.set_device(device)
        self.shard = _lib.Shard(hi - lo, device=device, id_base=lo)
        self.shard.set_codec(store.offset, store.scale)
        self._upload(store, lo, hi)
        cent, assign = make_list_major_resident(self.shard, nlist, centroids=cent, iters=int(ivf.get("iters", 10)),
                                                train_rows=ivf.get("train_rows"), seed=int(ivf.get("seed", 0)),
                                                offset=store.offset, scale=store.scale)
        self.ivf = {"nlist": nlist, "nprobe": min(int(ivf.get("nprobe", 256)), nlist), "centroids": cent,
                    "assign": assign.cpu().numpy()}

    def _upload(self, store, lo: int, hi: int, block_rows: int = 1 << 18):
        """rows [lo, hi) of the dump -> the shard.  Two pinned staging buffers of ``block_rows`` rows (192 MiB each): the
        HDF5 read of block t+1 overlaps the host->HBM copy of block t; host memory holds two blocks, not the dump."""
        import ctypes as C
        import torch
        if hi <= lo:
            return
        nbytes = block_rows * _lib.DIM
        ptrs, views = [], []
        for _ in range(2):
            p = C.c_void_p()
            _lib._chk(_lib.lib.dph_host_alloc_pinned(nbytes, C.byref(p)))
            ptrs.append(p)
            views.append(np.ctypeslib.as_array((C.c_int8 * nbytes).from_address(p.value)).reshape(block_rows, _lib.DIM))
        dev = torch.device("cuda", self.shard.device)
        stream = torch.cuda.Stream(device=dev)
        events = [None, None]
        try:
            turn, r0 = 0, lo
            starts = store.doc_starts()
            while r0 < hi:
                # the longest run of whole documents that fits a staging buffer (an oversized document goes alone)
                if r0 + block_rows >= hi:
                    r1 = hi
                else:
                    j = int(np.searchsorted(starts, r0 + block_rows, side="right")) - 1
                    r1 = int(starts[j])
                    if r1 <= r0:
                        k = int(np.searchsorted(starts, r0, side="right"))
                        r1 = min(int(starts[k]) if k < len(starts) else hi, hi)
                n = r1 - r0
                b = turn & 1
                if events[b] is not None:
                    events[b].synchronize()             # the upload that last used this buffer has finished
                if n <= block_rows:
                    self._read_rows(store, views[b][:n], r0)
                    self.shard.upload_async(ptrs[b].value, r0 - lo, n, stream.cuda_stream)
                    ev = torch.cuda.Event()
                    ev.record(stream)
                    events[b] = ev
                else:
                    tmp = np.empty((n, _lib.DIM), np.int8)
                    self._read_rows(store, tmp, r0)
                    self.shard.upload(tmp, r0 - lo)
                r0 = r1
                turn += 1
            stream.synchronize()
        finally:
            for p in ptrs:
                _lib.lib.dph_host_free_pinned(p)

    @staticmethod
    def _read_rows(store, out: np.ndarray, row0: int):
        if hasattr(store, "read_rows_into"):
            store.read_rows_into(out, row0)                 # ReferenceDump: HDF5 straight into the staging buffer
        else:
            out[:] = store.rows[row0:row0 + out.shape[0]]

    @classmethod
    def from_store(cls, store: DocStore, device: int = 0, logging_level=logging.WARNING, ivf: Optional[dict] = None):
        return cls(None, "in-memory", None, logging_level=logging_level, device=device, _store=store, ivf=ivf)

    @classmethod
    def from_shard(cls, shard: "_lib.Shard", store, logging_level=logging.WARNING):
        """Wrap a shard that is already resident and finalized (rows uploaded or generated on the device, idx2id and
        f2o set); ``store`` only has to answer ``doc_meta(doc_idx)``."""
        self = cls.__new__(cls)
        logger.setLevel(logging_level)
        self.phrase_dump_dir, self.index_path, self.max_idx = None, "resident-shard", int(1e8)
        self.cuda, self.num_docs_list = True, []
        self.store, self.shard = store, shard
        self.rank, self.world, self.dist = 0, 1, None
        self.force_collectives = False
        self.row_lo, self.row_hi = 0, shard.ntotal
        self.index = _IndexView(shard)
        self.R = np.eye(shard.d, dtype=np.float32)
        return self

    # ------------------------------------------------------------------ index.py:124-141
    def get_idxs(self, I):
        I = np.asarray(I, dtype=np.int64)
        if ((I < 0) | (I >= self.index.ntotal)).any() and getattr(self.store, "id_offsets", None) is None:
            logger.info("index out of range!")
        if self.world == 1 and not hasattr(self.store, "rows_of_ids"):
            doc, word = self.shard.id2docword(I)          # clips like the reference
            return doc.astype(np.int64), word.astype(np.int64)
        # whole-index lookup on the host (every rank holds idx2id, like the reference's load_idx_f): ids of a merged
        # index decode as offset + local (index.py:135-140); unknown ids clip to the ends (:133)
        if hasattr(self.store, "rows_of_ids"):
            rows = self.store.rows_of_ids(I)
            first = int(self.store.id_offsets[0]) if len(self.store.id_offsets) else 0
            rows = np.where(rows >= 0, rows, np.where(I < first, 0, self.store.n_rows - 1))
        else:
            rows = np.clip(I, 0, max(self.store.n_rows - 1, 0))
        return self.store.row2doc[rows].astype(np.int64), self.store.row2word[rows].astype(np.int64)

    # ------------------------------------------------------------------ index.py:167-187
    @staticmethod
    def adjust(each, delimiter=" [PAR] "):
        ctx = each["context"]
        lo = ctx.rfind(delimiter, 0, each["start_pos"])
        lo = 0 if lo == -1 else lo + len(delimiter)
        hi = ctx.find(delimiter, each["end_pos"])
        hi = len(ctx) if hi == -1 else hi
        each["context"] = ctx[lo:hi]
        each["start_pos"] -= lo
        each["end_pos"] -= lo
        return each

    @staticmethod
    def adjust_sent(each):
        sents = split_sentences(each["context"])
        starts = np.array([s for _, s in sents])
        a = int((starts <= each["start_pos"]).sum()) - 1
        b = int((starts <= each["end_pos"] - 1).sum()) - 1
        lo, hi = min(a, b), max(a, b)
        each["context"] = " ".join(sents[i][0] for i in range(lo, hi + 1))
        each["start_pos"] -= sents[lo][1]
        each["end_pos"] -= sents[lo][1]
        return each

    def _set_nprobe(self, nprobe):
        """index.py:52-62 sets ``nprobe`` on the IVF index once; the callers then pass ``nprobe=256`` with every search
        (model.py:82-87, eval_phrase_retrieval.py:72-77) and the reference ignores it (:191).  Here the argument is
        honoured PER CALL on a list-major / PQ shard (a flat shard has no lists: the search is exact whatever it says):
        the tuning key is set for the call and the configured value (``ivf={"nprobe": ...}``, 256 for a PQ file) comes back
        afterwards.  Every entry point -- search, search_dense, search_device, search_stream -- defaults to ``nprobe=None`` = the
        configured value (the reference's signature default of 256 is what its callers pass explicitly anyway; a silent 256 here
        would make ``MIPS(ivf={"nprobe": 64}).search(q)`` and ``.search_device(q)`` answer differently).  Returns the value to
        restore (None: nothing was changed)."""
        cfg = getattr(self, "ivf", None) or getattr(self, "pq", None)
        if cfg is None or nprobe is None:
            return None
        np_ = max(1, min(int(nprobe), cfg["nlist"]))
        if np_ == cfg["nprobe"]:
            return None
        prev = cfg["nprobe"]
        self.shard.set_tuning("nprobe", np_)
        cfg["nprobe"] = np_
        return prev

    # ------------------------------------------------------------------ index.py:189-218
    def search_dense(self, query, q_texts=None, nprobe=None, top_k=10):
        batch_size = query.shape[0]
        t0 = time()
        q = np.asarray(query).astype(np.float32)
        stacked = np.concatenate(np.split(q, 2, axis=1), axis=0)              # [2B, 768]: starts then ends
        prev = self._set_nprobe(nprobe)
        try:
            scores, I = self.index.search(stacked, top_k)
        finally:
            self._set_nprobe(prev)
        start_scores, start_I = scores[:batch_size], I[:batch_size]
        end_scores, end_I = scores[batch_size:], I[batch_size:]
        logger.debug(f"1) {time() - t0:.3f}s: MIPS")
        t0 = time()
        start_doc, start_word = self.get_idxs(start_I)
        end_doc, end_word = self.get_idxs(end_I)
        self.num_docs_list.append(sum(len(set(s.tolist() + e.tolist())) for s, e in zip(start_doc, end_doc)) / batch_size)
        logger.debug(f"2) {time() - t0:.3f}s: get index")
        return start_doc, start_word, start_I, end_doc, end_word, end_I, start_scores, end_scores

    # ------------------------------------------------------------------ index.py:220-422
    def search_phrase(self, query, start_doc_idxs, start_idxs, orig_start_idxs, end_doc_idxs, end_idxs, orig_end_idxs,
                      start_scores, end_scores, top_k=10, max_answer_length=10, return_idxs=False, return_sent=False):
        num_queries = query.shape[0]
        L = int(max_answer_length)
        q = np.asarray(query).astype(np.float32)
        q_start, q_end = np.split(q, 2, axis=1)
        flat = lambda a: np.reshape(np.asarray(a), [-1])                      # noqa: E731
        sdoc, sword, edoc, eword = flat(start_doc_idxs), flat(start_idxs), flat(end_doc_idxs), flat(end_idxs)
        sI, eI = flat(orig_start_idxs), flat(orig_end_idxs)
        sD, eD = flat(start_scores), flat(end_scores)
        assert len(sdoc) == len(sword) == len(eword) == len(sD) == num_queries * top_k

        t0 = time()
        # "find end for start": windows word+i over consecutive ids, dotted with the END half of the query
        pred_end, best1, _, v1 = self.shard.rescore(0, q_end, top_k, L, sI, sdoc, sword, sD, want_vecs=return_idxs)
        # "find start for end": windows word-i, dotted with the START half
        pred_start, best2, _, v2 = self.shard.rescore(1, q_start, top_k, L, eI, edoc, eword, eD, want_vecs=return_idxs)
        logger.debug(f"2,3) {time() - t0:.3f}s: find end / find start")

        v1a = (v1[:, 0, :], v1[:, 1, :]) if return_idxs else None
        v2a = (v2[:, 0, :], v2[:, 1, :]) if return_idxs else None
        return self._assemble(num_queries, top_k, sdoc, sword, edoc, eword, pred_end, best1, pred_start, best2,
                              v1a, v2a, return_sent)

    def _assemble(self, num_queries, top_k, sdoc, sword, edoc, eword, pred_end, best1, pred_start, best2, v1, v2,
                  return_sent=False, agg_strat=None):
        """The host half of search_phrase (index.py:373-421): interleave start/end candidates, metadata lookup, dict
        assembly, answer slice, paragraph / sentence cropping, per-query sort and dummy filter -- in C++
        (csrc/dph_host.cpp: one call per batch; ``_assemble_py`` below is the same thing in python and is what the
        tests hold it against).  ``agg_strat`` != None also runs ``aggregate_results`` (index.py:424-448) on every query's
        list inside the same call."""
        t0 = time()
        doc_i = np.stack([sdoc, edoc], 1).reshape(-1).astype(np.int64)     # (start-cand, end-cand) interleaved
        start_i = np.stack([sword, np.asarray(pred_start).astype(np.int64)], 1).reshape(-1).astype(np.int64)
        end_i = np.stack([np.asarray(pred_end).astype(np.int64), eword], 1).reshape(-1).astype(np.int64)
        score_i = np.stack([best1, best2], 1).reshape(-1).astype(np.float64)
        start_vecs = end_vecs = None
        if v1 is not None:
            start_vecs = np.stack([v1[0], v2[1]], 1).reshape(-1, v1[0].shape[-1])
            end_vecs = np.stack([v1[1], v2[0]], 1).reshape(-1, v1[0].shape[-1])
        host = self._host_half()
        if agg_strat is not None and agg_strat not in ("opt1", "opt2", "opt3", "opt4"):
            raise NotImplementedError("wrong aggregation strategy")
        out = host.assemble(int(num_queries), int(top_k), doc_i, start_i, end_i, score_i, start_vecs, end_vecs,
                            bool(return_sent), agg_strat, normalize_answer if agg_strat is not None else None)
        logger.debug(f"4) {time() - t0:.3f}s: get metadata")
        return out

    def _assemble_py(self, num_queries, top_k, sdoc, sword, edoc, eword, pred_end, best1, pred_start, best2, v1, v2,
                     return_sent=False):
        """``_assemble`` in python, line by line after index.py:373-421 (test reference for the C++ implementation)."""
        return_idxs = v1 is not None
        doc_i = np.stack([sdoc, edoc], 1).reshape(-1)                        # (start-cand, end-cand) interleaved
        start_i = np.stack([sword, pred_start.astype(np.int64)], 1).reshape(-1)
        end_i = np.stack([pred_end.astype(np.int64), eword], 1).reshape(-1)
        score_i = np.stack([best1, best2], 1).reshape(-1)
        if return_idxs:
            start_vecs = np.stack([v1[0], v2[1]], 1).reshape(-1, v1[0].shape[-1])
            end_vecs = np.stack([v1[1], v2[0]], 1).reshape(-1, v1[0].shape[-1])

        meta = {int(d): self.store.doc_meta(int(d)) for d in set(doc_i.tolist()) if d >= 0}
        out = []
        for g, (d, s, e, sc) in enumerate(zip(doc_i.tolist(), start_i.tolist(), end_i.tolist(), score_i.tolist())):
            if d < 0:
                out.append({"score": _DUMMY_SCORE, "context": "dummy", "start_pos": 0, "end_pos": 0, "title": [""]})
                continue
            m = meta[d]
            start_pos = int(m.word2char_start[m.f2o_start[s]])
            if len(m.word2char_end) > 0 and e >= 0:
                end_pos = int(m.word2char_end[m.f2o_start[e]])
            else:
                end_pos = start_pos + 1
            out.append({
                "context": m.context, "title": [m.title], "doc_idx": d, "start_pos": start_pos, "end_pos": end_pos,
                "start_idx": s, "end_idx": e, "score": sc,
                "start_vec": start_vecs[g] if return_idxs else None,
                "end_vec": end_vecs[g] if return_idxs else None,
            })
        for each in out:
            each["answer"] = each["context"][each["start_pos"]:each["end_pos"]]
        out = [self.adjust(each) for each in out]
        if return_sent:
            out = [self.adjust_sent(each) for each in out]

        new_out = [[] for _ in range(num_queries)]
        per_q = 2 * top_k
        for g, each in enumerate(out):
            new_out[g // per_q].append(each)
        for i in range(num_queries):
            new_out[i] = sorted(new_out[i], key=lambda r: -r["score"])
            new_out[i] = [r for r in new_out[i] if r["score"] > _DROP_BELOW]
        return new_out

    # ------------------------------------------------------------------ index.py:424-448
    def aggregate_results(self, results, top_k=10, q_text=None, agg_strat="opt1"):
        """index.py:424-448 in C++ (csrc/dph_host.cpp: aggregate); ``_aggregate_results_py`` is the python restatement
        the tests hold it against."""
        if agg_strat not in ("opt1", "opt2", "opt3", "opt4"):
            raise NotImplementedError("wrong aggregation strategy")
        return _dph_host.aggregate(results, agg_strat, normalize_answer)

    @staticmethod
    def _aggregate_results_py(results, top_k=10, q_text=None, agg_strat="opt1"):
        first: Dict[str, int] = {}
        for r_idx, r in enumerate(results):
            if agg_strat == "opt1":
                key = f'{r["title"]}_{r["start_pos"]}_{r["end_pos"]}'
            elif agg_strat == "opt2":
                key = f'{r["context"]}'
            elif agg_strat == "opt3":
                key = f'{r["title"]}'
            elif agg_strat == "opt4":
                key = f'{normalize_answer(r["answer"])}'
            else:
                raise NotImplementedError("wrong aggregation strategy")
            if key not in first:
                first[key] = r_idx
            else:
                r["score"] = _DUMMY_SCORE
                if agg_strat == "opt4" and r["title"][0] not in results[first[key]]["title"]:
                    results[first[key]]["title"] += r["title"]
        results = sorted(results, key=lambda r: -r["score"])
        return [r for r in results if r["score"] > _DROP_BELOW]

    # ------------------------------------------------------------------ index.py:450-482
    def search(self, query, q_texts=None, nprobe=None, top_k=10, aggregate=False, return_idxs=False,
               max_answer_length=10, agg_strat="opt1", return_sent=False):
        # range-sharded (world > 1) this is a collective: every rank calls it with the same batch and gets the same merged result
        # one GPU: the same device-resident chain the streaming forms use -- ONE upload of the query batch, search + both
        # window passes enqueued back to back, ONE download of the packed [2B, k] record, then the C++ host half -- instead
        # of the reference's two stages with a host round trip (and a scratch allocation) each: search_dense + search_phrase
        # stay available and return the same things (the goldens hold both)
        t0 = time()
        L = int(max_answer_length)
        prev = self._set_nprobe(nprobe)
        try:
            outs = self._finish(self._enqueue(query, top_k, L, 0), top_k, L, return_sent, aggregate, agg_strat, q_texts,
                                return_idxs=return_idxs)
        finally:
            self._set_nprobe(prev)
        logger.debug(f"Top-{top_k} MIPS + phrase search: {time() - t0:.3f}s")
        return outs

    # ------------------------------------------------------------------ device-resident / streaming forms
    # SURVEY.md 8(f) rank 4 (keep the encoder's query vectors on the device; drop the .tolist() round trip of
    # open_utils.py:97) and 8(d) config 5 (continuous batches): the GPU half of batch t+1 is enqueued before the
    # host half (metadata, dicts, cropping, de-duplication) of batch t runs, so the two overlap.
    def _searcher(self, B, k, L, slot):
        from .dist import ShardedSearcher
        import torch
        if not hasattr(self, "_searchers"):
            self._searchers = {}
        key = (B, k, L, slot)
        if key not in self._searchers:
            dev = torch.device("cuda", self.shard.device)
            ss = ShardedSearcher(self.shard, B, k, L, rank=self.rank, world=self.world, dist=self.dist, device=dev,
                                 force_collectives=self.force_collectives)
            ss.host = torch.empty(ss.layout.nbytes, dtype=torch.uint8).pin_memory()
            # host queries reach the device through a pinned staging buffer of the slot, asynchronously: a pageable `.to(device)`
            # BLOCKS the host until the copy has run -- behind the previous batch's kernels on the same stream, i.e. the host half
            # of batch t could not start before the GPU half of batch t had finished (1.3 ms of a 9.1 ms batch of 512 over PQ)
            ss.qhost = torch.empty((B, 2 * self.shard.d), dtype=torch.float32).pin_memory()
            ss.qdev = torch.empty((B, 2 * self.shard.d), dtype=torch.float32, device=dev)
            ss.done = torch.cuda.Event()
            self._searchers[key] = ss
        return self._searchers[key]

    def _enqueue(self, query, top_k, L, slot):
        """Asynchronous GPU half of one batch: search + both window passes + device->pinned-host copy of the record."""
        import torch
        dev = torch.device("cuda", self.shard.device)
        ss = None
        if isinstance(query, torch.Tensor):
            q = query.detach().to(device=dev, dtype=torch.float32)
        else:
            qn = np.asarray(query)
            if qn.ndim != 2 or qn.shape[1] != 2 * self.shard.d:
                raise ValueError(f"query must be [B, {2 * self.shard.d}]")
            ss = self._searcher(qn.shape[0], top_k, L, slot)
            # (the slot's staging buffers are free: the batch that used them two steps ago has been finished by the caller)
            np.copyto(ss.qhost.numpy(), qn, casting="unsafe")
            with torch.cuda.device(dev):
                q = ss.qdev.copy_(ss.qhost, non_blocking=True)
        if q.dim() != 2 or q.shape[1] != 2 * self.shard.d:
            raise ValueError(f"query must be [B, {2 * self.shard.d}]")
        B = q.shape[0]
        ss = ss or self._searcher(B, top_k, L, slot)
        t0 = time()
        with torch.cuda.device(dev):
            ss.step(q.contiguous())
            ss.host.copy_(ss.result_record, non_blocking=True)
            ss.done.record()
        self._timing()["enqueue_s"] += time() - t0
        return ss, q

    def _timing(self):
        """Where the wall time of the device-resident forms went, summed since the last ``reset_timing()``: enqueueing the GPU
        half, waiting for its record (GPU-bound time), the host half (metadata, dicts, cropping, de-duplication)."""
        return self.__dict__.setdefault("timing", {"enqueue_s": 0.0, "wait_s": 0.0, "host_s": 0.0, "host_idx_s": 0.0, "host_assemble_s": 0.0, "batches": 0})

    def reset_timing(self):
        self.__dict__.pop("timing", None)
        return self._timing()

    def _finish(self, pending, top_k, L, return_sent, aggregate, agg_strat, q_texts, return_idxs=False):
        """Host half: wait for the record, repair uncertified rows through the exact host chain, assemble the dicts."""
        ss, q = pending
        t0 = time()
        ss.done.synchronize()
        t1 = time()
        tm = self._timing()
        tm["wait_s"] += t1 - t0                   # the host had nothing to do: the GPU half of this batch was still running
        tm["batches"] += 1
        try:
            return self._finish_host(ss, q, top_k, L, return_sent, aggregate, agg_strat, q_texts, return_idxs)
        finally:
            tm["host_s"] += time() - t1

    def _finish_host(self, ss, q, top_k, L, return_sent, aggregate, agg_strat, q_texts, return_idxs):
        B = ss.B
        v = ss.layout.views(ss.host)
        D, I = v["D"].numpy(), v["I"].numpy()
        best, pred, status = v["best"].numpy(), v["pred"].numpy(), v["status"].numpy()
        if (status == 1).any():                     # rare: libdph already retried on the device (dist.step_exact); 3 = a non-finite query row, final
            out = ss.step_exact(q)
            D, I = out["D"].cpu().numpy(), out["I"].cpu().numpy()
            best, pred = out["best"].cpu().numpy(), out["pred"].cpu().numpy()
        t_a = time()
        sdoc, sword = self.get_idxs(I[:B])
        edoc, eword = self.get_idxs(I[B:])
        # distinct documents per query among its 2 * top_k candidates (index.py:213-214 keeps the mean per batch): sorted rows, counted
        # where neighbours differ -- a python set per query was 1.5 ms of a batch of 512
        both = np.sort(np.concatenate([np.reshape(sdoc, [B, -1]), np.reshape(edoc, [B, -1])], 1), 1)
        self.num_docs_list.append(float(((both[:, 1:] != both[:, :-1]).sum(1) + 1).sum()) / max(B, 1) if both.size else 0.0)
        flat = lambda a: np.reshape(a, [-1])          # noqa: E731
        v1 = v2 = None
        if return_idxs:
            v1, v2 = self._window_vectors(q, top_k, L, D, I, sdoc, sword, edoc, eword)
        t_b = time()
        out = self._assemble(B, top_k, flat(sdoc), flat(sword), flat(edoc), flat(eword), flat(pred[:B]),
                             flat(best[:B]), flat(pred[B:]), flat(best[B:]), v1, v2, return_sent,
                             agg_strat=agg_strat if aggregate else None)
        tm = self._timing()                          # (where the host half goes: id -> (doc, word) look-ups, then metadata / dicts / de-duplication)
        tm["host_idx_s"] += t_b - t_a
        tm["host_assemble_s"] += time() - t_b
        return out

    def _window_vectors(self, q, top_k, L, D, I, sdoc, sword, edoc, eword):
        """start/end vectors of the merged candidates for ``return_idxs`` (index.py:381-389): every rank re-runs the
        window kernel on the merged ids -- rows it does not hold come back as zero vectors (the reference's reconstruct
        failure path, :285-288) -- and a SUM all-reduce assembles them: exactly one rank contributes each vector."""
        import torch
        B = q.shape[0]
        qn = q.detach().cpu().numpy()
        q_start, q_end = qn[:, :768], qn[:, 768:]
        flat = lambda a: np.reshape(np.asarray(a), [-1])          # noqa: E731
        _, _, _, a = self.shard.rescore(0, q_end, top_k, L, flat(I[:B]), flat(sdoc), flat(sword), flat(D[:B]), want_vecs=True)
        _, _, _, b = self.shard.rescore(1, q_start, top_k, L, flat(I[B:]), flat(edoc), flat(eword), flat(D[B:]), want_vecs=True)
        if self.world > 1 or self.force_collectives:
            dev = torch.device("cuda", self.shard.device)
            t = torch.from_numpy(np.stack([a, b])).to(dev)
            self.dist.all_reduce(t)
            a, b = t[0].cpu().numpy(), t[1].cpu().numpy()
        return (a[:, 0, :], a[:, 1, :]), (b[:, 0, :], b[:, 1, :])

    def search_device(self, query, q_texts=None, top_k=10, aggregate=False, max_answer_length=10, agg_strat="opt1",
                      return_sent=False, nprobe=None):
        """``search`` for a query batch that already lives on the GPU (a torch tensor straight from the encoder):
        nothing but the [2B, k] result record crosses PCIe.  Same results as ``search`` under the same ``nprobe``
        (None = the configured one: ``ivf={"nprobe": ...}`` / 256 for a PQ file)."""
        L = int(max_answer_length)
        prev = self._set_nprobe(nprobe)
        try:
            return self._finish(self._enqueue(query, top_k, L, 0), top_k, L, return_sent, aggregate, agg_strat, q_texts)
        finally:
            self._set_nprobe(prev)

    def _host_half(self):
        host = getattr(self, "_host", None)
        if host is None or getattr(self, "_host_store", None) is not self.store:
            # (the callback closes over the STORE, not over self: HostHalf is an extension type the cycle collector cannot look into, so
            # a callback that held this MIPS would keep it -- and the shard's HBM -- alive for the life of the process)
            store = self.store
            host = self._host = _dph_host.HostHalf(lambda d: store.doc_meta(int(d)))
            self._host_store = store
        return host

    def close(self):
        """Release the shard's device memory now (otherwise: when the object is collected).  The reference's MIPS has no counterpart
        (its FAISS index lives as long as the process); a host that swaps indexes needs one."""
        self._host = None
        self._host_store = None
        if getattr(self, "_searchers", None):
            self._searchers.clear()                  # (per-shape search state: device and pinned buffers of its own)
        shard = getattr(self, "shard", None)
        if shard is not None:
            shard.close()

    def _prepare_async(self, pending, top_k, return_sent, agg):
        """Hand the record of an enqueued batch to the host half's worker thread (csrc/dph_host.cpp prepare_async): ids -> (doc, word),
        interleaving, document cache, answer / paragraph / sentence positions, per-query sort and de-duplication -- none of it needs
        the interpreter, so it runs while this thread turns the PREVIOUS batch into result dicts."""
        import ctypes
        ss, _ = pending
        t0 = time()
        ss.done.synchronize()
        tm = self._timing()
        tm["wait_s"] += time() - t0
        tm["batches"] += 1
        v = ss.layout.views(ss.host)
        fn = ctypes.cast(_lib.lib.dph_id2docword, ctypes.c_void_p).value
        return self._host_half().prepare_async(ss.B, int(top_k), v["I"].data_ptr(), v["best"].data_ptr(), v["pred"].data_ptr(), v["status"].data_ptr(),
                                               fn, self.shard._h.value, bool(return_sent), agg)

    def search_stream(self, batches, q_texts=None, top_k=10, aggregate=False, max_answer_length=10, agg_strat="opt1",
                      return_sent=False, nprobe=None):
        """Generator over an iterable of query batches (numpy or device tensors, [B, 1536]); yields what ``search``
        would return for each, in order, while the GPU already works on the next batch (two record slots).
        Single-rank indexes run THREE batches deep (round 6): the GPU half of batch t, the interpreter-free part of the host half of
        batch t-1 on a worker thread, and the result dicts of batch t-2 on this thread -- over a PQ index at batch 512 the host half
        (7 ms, most of it CPython object creation) used to be all there was to wait for."""
        L = int(max_answer_length)
        texts = iter(q_texts) if q_texts is not None else None
        if aggregate and agg_strat not in ("opt1", "opt2", "opt3", "opt4"):
            raise NotImplementedError("wrong aggregation strategy")
        agg = agg_strat if aggregate else None
        deep = (self.world == 1 and not self.force_collectives and not hasattr(self.store, "rows_of_ids")
                and os.environ.get("DPH_STREAM_DEPTH", "") != "2")
        # ... for LARGE batches: a worker thread per batch costs ~0.1 ms, which a batch of 64 (0.6 ms of host half) does not earn back
        # (measured over the PQ index: 58 k Q/s three deep against 65 k two deep); DPH_STREAM_DEPTH=3 forces it
        import itertools
        batches = iter(batches)
        first = next(batches, None)
        if first is None:
            return
        if os.environ.get("DPH_STREAM_DEPTH", "") != "3" and 2 * len(first) * int(top_k) < 8192:
            deep = False
        batches = itertools.chain([first], batches)
        restore = self._set_nprobe(nprobe)             # for the whole stream (the generator restores it when it ends)
        try:
            if not deep:
                prev, prev_t, t = None, None, 0
                for q in batches:
                    cur = self._enqueue(q, top_k, L, t & 1)
                    cur_t = next(texts) if texts is not None else None
                    if prev is not None:
                        yield self._finish(prev, top_k, L, return_sent, aggregate, agg_strat, prev_t)
                    prev, prev_t, t = cur, cur_t, t + 1
                if prev is not None:
                    yield self._finish(prev, top_k, L, return_sent, aggregate, agg_strat, prev_t)
                return
            host = self._host_half()
            norm = normalize_answer if agg is not None else None

            def collect(item):
                """(Prepared, pending, texts) -> the batch's result lists; the rare batch with an uncertified row is repaired the
                synchronous way -- its slot's buffers are still its own: the caller collects BEFORE it re-uses the slot"""
                P, pending, p_texts = item
                t0 = time()
                try:
                    needs_exact, num_docs = P.wait()
                    if needs_exact:
                        ss, q = pending
                        return self._finish_host(ss, q, top_k, L, return_sent, aggregate, agg_strat, p_texts, False)
                    self.num_docs_list.append(float(num_docs))
                    return P
                finally:
                    self._timing()["host_s"] += time() - t0

            def finish(ready):
                if not isinstance(ready, _dph_host.Prepared):
                    return ready
                t0 = time()
                try:
                    return host.materialize(ready, norm)
                finally:
                    tm = self._timing()
                    tm["host_s"] += time() - t0
                    tm["host_assemble_s"] += time() - t0

            pend, prep, t = None, None, 0      # pend: GPU half enqueued; prep: host half being prepared on the worker
            for q in batches:
                cur_t = next(texts) if texts is not None else None
                ready = collect(prep) if prep is not None else None          # batch t-2 is prepared: its slot may be re-used now
                cur = self._enqueue(q, top_k, L, t & 1)
                prep = (self._prepare_async(pend[0], top_k, return_sent, agg), pend[0], pend[1]) if pend is not None else None
                if ready is not None:
                    yield finish(ready)
                pend, t = (cur, cur_t), t + 1
            if prep is not None:
                yield finish(collect(prep))
            if pend is not None:
                yield finish(collect((self._prepare_async(pend[0], top_k, return_sent, agg), pend[0], pend[1])))
        finally:
            self._set_nprobe(restore)
