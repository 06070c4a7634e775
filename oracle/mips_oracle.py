"""CPU oracle for the DensePhrases phrase-retrieval hot path (MIPS.search).

TEST INFRASTRUCTURE ONLY.  Nothing under ``densephrases_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker / timed CPU comparator.

What it restates (all citations relative to /root/reference):

* int8 phrase-vector codec ............ densephrases/utils/embed_utils.py:141-149
* FAISS ``IndexFlatIP.search`` ......... call site densephrases/index.py:200
  (faiss-gpu==1.6.5, requirements.txt:2 -- NOT vendored, NOT installed here;
  its published semantics are restated: D = k largest <x_i, y_j> sorted
  descending, I = -1 / D = -FLT_MAX padding when fewer than k rows exist.
  FAISS 1.6.5 leaves the order of exactly tied scores unspecified; the oracle
  fixes it to (score desc, id asc).)
* id -> (doc, word) mapping ............ densephrases/index.py:124-141
* query stacking / output tuple ........ densephrases/index.py:189-218
* window re-scoring, masks, interleave . densephrases/index.py:220-422
* paragraph crop ....................... densephrases/index.py:167-176
* result aggregation opt1..opt4 ........ densephrases/index.py:424-448
* orchestration ........................ densephrases/index.py:450-482

Parity pin status: the index.py logic (everything except the FAISS arithmetic)
is pinned against the reference's own code, executed unmodified from
/root/reference with stand-in modules for its missing third-party imports
(oracle/refshim/, oracle/make_golden.py -> tests/golden/*.npz|json).  The FAISS
inner-product arithmetic itself is *restated, not pinned*: FAISS is absent from
this container and the reference ships no golden vectors for it ("parity
unpinned" for that one function).
"""
from __future__ import annotations

import re
import string
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

FLT_MAX = np.float32(3.4028234663852886e38)
NEG_MASK = -1e9          # index.py:331,356
DUMMY_SCORE = -1e8       # index.py:400-401, 441
DROP_BELOW = -1e5        # index.py:420, 447


# --------------------------------------------------------------------------
# codec  (embed_utils.py:141-149)
# --------------------------------------------------------------------------
def float_to_int8(num: np.ndarray, offset: float = -2.0, factor: float = 20.0) -> np.ndarray:
    out = (np.asarray(num) - offset) * factor
    out = np.clip(out, -128, 127)
    return np.round(out).astype(np.int8)


def int8_to_float(num: np.ndarray, offset: float = -2.0, factor: float = 20.0) -> np.ndarray:
    # float32 array / python scalar + python scalar stays float32: two roundings.
    return num.astype(np.float32) / np.float32(factor) + np.float32(offset)


def dequant_lut(offset: float = -2.0, factor: float = 20.0) -> np.ndarray:
    """x32[n + 128] for n in [-128, 127]: the 256 fp32 values the reference feeds FAISS."""
    return int8_to_float(np.arange(-128, 128, dtype=np.int16).astype(np.int8), offset, factor)


# --------------------------------------------------------------------------
# FAISS IndexFlatIP.search restated  (call site index.py:200)
# --------------------------------------------------------------------------
def _canonical_topk(scores: np.ndarray, ids: np.ndarray, k: int):
    """rows of (score desc, id asc); scores float64 [n, m], ids int64 [m] or [n, m]."""
    n, m = scores.shape
    if ids.ndim == 1:
        ids = np.broadcast_to(ids, (n, m))
    kk = min(k, m)
    D = np.full((n, k), -np.inf, dtype=np.float64)
    I = np.full((n, k), -1, dtype=np.int64)
    for r in range(n):
        order = np.lexsort((ids[r], -scores[r]))[:kk]
        D[r, :kk] = scores[r, order]
        I[r, :kk] = ids[r, order]
    return D, I


def flat_ip_search(xq: np.ndarray, xb_int8: np.ndarray, k: int, offset: float = -2.0,
                   factor: float = 20.0, block: int = 65536, id_base: int = 0):
    """Exact inner-product top-k over the de-quantised dump.

    Scores are the float64 dot products of the fp32 query with the *fp32*
    de-quantised vectors x32 = fl(fl(n)/20 + (-2)) -- i.e. the infinitely
    precise value of the sgemm FAISS runs on the reference's inputs.  Returns
    (D float32 [n,k] descending, I int64 [n,k], D64 float64) with FAISS padding.
    """
    xq = np.ascontiguousarray(xq, dtype=np.float32)
    n, d = xq.shape
    N = xb_int8.shape[0]
    q64 = xq.astype(np.float64)
    best_s = np.empty((n, 0), dtype=np.float64)
    best_i = np.empty((n, 0), dtype=np.int64)
    for b0 in range(0, N, block):
        xb = int8_to_float(xb_int8[b0:b0 + block], offset, factor).astype(np.float64)
        s = q64 @ xb.T                                  # [n, blk] float64
        ids = np.arange(b0, b0 + xb.shape[0], dtype=np.int64) + id_base
        cand_s = np.concatenate([best_s, s], axis=1)
        cand_i = np.concatenate([best_i, np.broadcast_to(ids, s.shape)], axis=1)
        if cand_s.shape[1] > 4 * k + 64:
            # cheap pre-filter keeping every row that can still be in the top-k
            kth = np.partition(cand_s, cand_s.shape[1] - min(k, cand_s.shape[1]), axis=1)[
                :, cand_s.shape[1] - min(k, cand_s.shape[1])]
            keep = cand_s >= kth[:, None]
            width = int(keep.sum(1).max())
            ns = np.full((n, width), -np.inf)
            ni = np.full((n, width), np.iinfo(np.int64).max, dtype=np.int64)
            for r in range(n):
                sel = np.nonzero(keep[r])[0]
                ns[r, :sel.size] = cand_s[r, sel]
                ni[r, :sel.size] = cand_i[r, sel]
            cand_s, cand_i = ns, ni
        best_s, best_i = cand_s, cand_i
    if best_s.shape[1] == 0:
        D64 = np.full((n, k), -np.inf)
        I = np.full((n, k), -1, dtype=np.int64)
    else:
        D64, I = _canonical_topk(best_s, best_i, k)
        bad = ~np.isfinite(D64)
        I[bad] = -1
    D = np.where(I >= 0, D64, -np.float64(FLT_MAX)).astype(np.float32)
    return D, I, D64


def flat_ip_search_sgemm(xq: np.ndarray, xb_int8: np.ndarray, k: int, offset: float = -2.0,
                         factor: float = 20.0, block: int = 1024, use_torch: bool = True):
    """The FAISS-CPU execution shape for nq >= 20: fp32 de-quantised database
    blocks of 1024 rows, one sgemm per block, running top-k merge.  This is the
    timed CPU comparator (bench.py cpu_baseline); fp32 arithmetic, all cores."""
    xq = np.ascontiguousarray(xq, dtype=np.float32)
    n = xq.shape[0]
    N = xb_int8.shape[0]
    if use_torch:
        import torch
        tq = torch.from_numpy(xq)
        best_s = torch.full((n, k), -float("inf"))
        best_i = torch.full((n, k), -1, dtype=torch.int64)
        # group several 1024-row FAISS blocks per topk merge to keep python overhead out of the timing
        step = block * 64
        tb = torch.from_numpy(xb_int8)
        for b0 in range(0, N, step):
            xb = tb[b0:b0 + step].to(torch.float32) / factor + offset
            s = tq @ xb.T
            kk = min(k, s.shape[1])
            ts, ti = torch.topk(s, kk, dim=1)
            cs = torch.cat([best_s, ts], 1)
            ci = torch.cat([best_i, ti + b0], 1)
            o = torch.topk(cs, k, dim=1)
            best_s, best_i = o.values, torch.gather(ci, 1, o.indices)
        return best_s.numpy(), best_i.numpy()
    best_s = np.full((n, k), -np.inf, dtype=np.float32)
    best_i = np.full((n, k), -1, dtype=np.int64)
    for b0 in range(0, N, block):
        xb = int8_to_float(xb_int8[b0:b0 + block], offset, factor)
        s = xq @ xb.T
        cs = np.concatenate([best_s, s], 1)
        ci = np.concatenate([best_i, np.broadcast_to(np.arange(b0, b0 + xb.shape[0]), s.shape)], 1)
        o = np.argsort(-cs, axis=1, kind="stable")[:, :k]
        best_s = np.take_along_axis(cs, o, 1)
        best_i = np.take_along_axis(ci, o, 1)
    return best_s, best_i


def flat_ip_search_fp32_resident(xq: np.ndarray, xb_blocks, k: int, id_base: int = 0):
    """What FAISS-CPU IndexFlatIP does with its index in RAM, restated (numpy on the host BLAS, all cores): the
    database is RESIDENT as fp32 (a list of [rows, 768] float32 arrays, de-quantised once when the index was built --
    not per query), each block is ONE sgemm against the query batch -- in the operand order the BLAS runs fast
    (``block @ queries.T``: the same product FAISS asks sgemm for, 20x faster here than ``queries @ block.T``) -- and the
    running top-k is only touched for the query rows whose block maximum beats their current k-th score (FAISS'
    heap_addn does the same comparison per element).  Returns (D float32 [n,k], I int64 [n,k]).  This is the timed CPU
    comparator of bench.py (`cpu_baseline`, run by oracle/cpu_baseline.py in a process of its own: with torch loaded into
    the same process numpy's BLAS runs 6x slower here -- two OpenMP runtimes fighting over the cores)."""
    qT = np.ascontiguousarray(np.asarray(xq, dtype=np.float32).T)          # [768, n]
    n = qT.shape[1]
    best_s = np.full((n, k), -np.inf, dtype=np.float32)
    best_i = np.full((n, k), -1, dtype=np.int64)
    b0 = id_base
    for xb in xb_blocks:
        s = np.asarray(xb) @ qT                                             # [rows, n] one sgemm
        upd = np.nonzero(s.max(axis=0) > best_s[:, -1])[0]
        if upd.size:
            sub = np.ascontiguousarray(s[:, upd].T)                         # [u, rows]: only the rows that improve
            kk = min(k, sub.shape[1])
            ti = np.argpartition(-sub, kk - 1, axis=1)[:, :kk]
            cs = np.concatenate([best_s[upd], np.take_along_axis(sub, ti, 1)], 1)
            ci = np.concatenate([best_i[upd], ti.astype(np.int64) + b0], 1)
            o = np.argsort(-cs, axis=1, kind="stable")[:, :k]
            best_s[upd] = np.take_along_axis(cs, o, 1)
            best_i[upd] = np.take_along_axis(ci, o, 1)
        b0 += xb.shape[0]
    return best_s, best_i


def ivf_flat_search(xq, xb_int8, centroids, assign, nprobe, k, offset=-2.0, factor=20.0):
    """IVF with exact in-list inner product (FAISS IndexIVFFlat semantics restated):
    coarse = flat IP over centroids, top-nprobe lists per query row, scan only
    those lists, global top-k in (score desc, id asc) order."""
    xq = np.ascontiguousarray(xq, dtype=np.float32)
    n = xq.shape[0]
    cs = xq.astype(np.float64) @ centroids.astype(np.float64).T
    _, probe = _canonical_topk(cs, np.arange(centroids.shape[0], dtype=np.int64), nprobe)
    D = np.full((n, k), -np.inf)
    I = np.full((n, k), -1, dtype=np.int64)
    lists = [np.nonzero(assign == c)[0] for c in range(centroids.shape[0])]
    for r in range(n):
        rows = np.concatenate([lists[c] for c in probe[r] if c >= 0]) if nprobe else np.empty(0, np.int64)
        if rows.size == 0:
            continue
        x = int8_to_float(xb_int8[rows], offset, factor).astype(np.float64)
        s = x @ xq[r].astype(np.float64)
        d1, i1 = _canonical_topk(s[None, :], rows.astype(np.int64), k)
        D[r], I[r] = d1[0], i1[0]
    I[~np.isfinite(D)] = -1
    D32 = np.where(I >= 0, D, -np.float64(FLT_MAX)).astype(np.float32)
    return D32, I, D


# --------------------------------------------------------------------------
# tie-aware comparison helper used by the parity tests
# --------------------------------------------------------------------------
def topk_equivalent(D_a, I_a, D_ref64, I_ref, rtol=2e-6, atol=1e-5):
    """True iff every row of (D_a, I_a) is the reference top-k up to swaps among
    scores closer than the fp32 evaluation noise of an sgemm.  Returns (ok, message)."""
    D_a = np.asarray(D_a, dtype=np.float64)
    I_a = np.asarray(I_a)
    n, k = I_ref.shape
    for r in range(n):
        valid = I_ref[r] >= 0
        if not np.array_equal(I_a[r] >= 0, valid):
            return False, f"row {r}: padding differs {I_a[r]} vs {I_ref[r]}"
        if not valid.any():
            continue
        if not np.allclose(D_a[r][valid], D_ref64[r][valid], rtol=rtol, atol=atol):
            return False, f"row {r}: scores differ {D_a[r]} vs {D_ref64[r]}"
        if np.array_equal(I_a[r], I_ref[r]):
            continue
        tol = atol + rtol * np.abs(D_ref64[r][valid]).max()
        kth = D_ref64[r][valid][-1]
        ref_pos = {int(i): j for j, i in enumerate(I_ref[r]) if i >= 0}
        a_ids = set(int(i) for i in I_a[r] if i >= 0)
        if len(a_ids) != int(valid.sum()):
            return False, f"row {r}: duplicate ids {I_a[r]}"
        for i, j in ref_pos.items():
            if i not in a_ids and D_ref64[r][j] > kth + tol:
                return False, f"row {r}: id {i} (score {D_ref64[r][j]}) missing"
        for j, i in enumerate(I_a[r]):
            i = int(i)
            if i < 0 or i == I_ref[r][j]:
                continue
            if i in ref_pos:
                if abs(D_ref64[r][ref_pos[i]] - D_ref64[r][j]) > tol:
                    return False, f"row {r} col {j}: id {i} out of order beyond tolerance"
            elif D_a[r][j] < kth - tol:
                return False, f"row {r} col {j}: id {i} is not a boundary near-tie"
    return True, "ok"


# --------------------------------------------------------------------------
# dump / index containers used by the restated index.py logic
# --------------------------------------------------------------------------
@dataclass
class DocMeta:
    """One ``/<doc_idx>`` group of phrase/*.hdf5 (embed_utils.py:235-246)."""
    doc_idx: int
    title: str
    context: str
    f2o_start: np.ndarray            # int64 [n_f]
    word2char_start: np.ndarray      # int32 [n_tok]
    word2char_end: np.ndarray        # int32 [n_tok]
    start: np.ndarray                # int8 [n_f, d]
    offset: float = -2.0
    scale: float = 20.0


@dataclass
class OracleIndex:
    """A flat (R = I) index over a dump: rows in idx2id order
    (build_phrase_index.py:192-276)."""
    xb: np.ndarray                   # int8 [N, d]
    row2doc: np.ndarray              # int32 [N]   idx2id 'doc'
    row2word: np.ndarray             # int32 [N]   idx2id 'word'
    docs: Dict[int, DocMeta]
    max_idx: int = int(1e8)          # index.py:33
    offset_groups: Dict[int, slice] = field(default_factory=dict)

    @property
    def ntotal(self) -> int:
        return int(self.xb.shape[0])

    @property
    def d(self) -> int:
        return int(self.xb.shape[1])


def build_index_from_docs(docs: Sequence[DocMeta]) -> OracleIndex:
    """Row order = iteration order of the dump's groups (h5py: string-sorted keys)
    with empty docs skipped (build_phrase_index.py:196-252)."""
    ordered = sorted(docs, key=lambda m: str(m.doc_idx))
    ordered = [m for m in ordered if m.start.shape[0] > 0]
    xb = np.concatenate([m.start for m in ordered], 0)
    row2doc = np.concatenate([np.full(m.start.shape[0], m.doc_idx, np.int32) for m in ordered])
    row2word = np.concatenate([np.arange(m.start.shape[0], dtype=np.int32) for m in ordered])
    return OracleIndex(xb=xb, row2doc=row2doc, row2word=row2word,
                       docs={m.doc_idx: m for m in docs})


# --------------------------------------------------------------------------
# index.py:124-141
# --------------------------------------------------------------------------
def get_idxs(index: OracleIndex, I: np.ndarray):
    I = np.asarray(I)
    if ((I < 0) | (I >= index.ntotal)).any():
        I = np.clip(I, 0, index.ntotal - 1)
    idxs = I % int(index.max_idx)        # single offset group "0" in a flat toy index
    return index.row2doc[idxs].astype(np.int64), index.row2word[idxs].astype(np.int64)


# --------------------------------------------------------------------------
# index.py:189-218
# --------------------------------------------------------------------------
def search_dense(index: OracleIndex, query: np.ndarray, top_k: int):
    B = query.shape[0]
    q = query.astype(np.float32)
    qs, qe = np.split(q, 2, axis=1)
    stacked = np.concatenate([qs, qe], 0)
    D, I, _ = flat_ip_search(stacked, index.xb, top_k)
    sD, sI, eD, eI = D[:B], I[:B], D[B:], I[B:]
    sdoc, sword = get_idxs(index, sI)
    edoc, eword = get_idxs(index, eI)
    return sdoc, sword, sI, edoc, eword, eI, sD, eD


# --------------------------------------------------------------------------
# index.py:305-321
# --------------------------------------------------------------------------
def _valid_phrase(index: OracleIndex, s: int, e: int, doc: int, L: int) -> bool:
    if doc < 0:
        return False
    f2o = index.docs[doc].f2o_start
    if s < 0 or s >= len(f2o) or e < 0 or e >= len(f2o):
        return False
    gap = int(f2o[e]) - int(f2o[s])
    return 0 <= gap <= L


def window_rescore(index: OracleIndex, query: np.ndarray, doc: np.ndarray, word: np.ndarray,
                   ids: np.ndarray, first_scores: np.ndarray, L: int, direction: str,
                   branch: str = "ram"):
    """One half of index.py:323-370.

    direction 'end'  : candidates are starts, window = word+i, i in [0,L)        (:323-346)
    direction 'start': candidates are ends,   window = word-i, i = L-1..0,
                       right-aligned                                             (:348-371)
    branch 'ram'  : rows fetched by consecutive *global ids* (index.py:282-300,
                    reconstruct failure -> zeros); 'hdf5': rows fetched by
                    within-doc word index, zero padded (index.py:263-272, 332-336).
    query: [Q, d] the half of the query this direction dots with
           (query_end for 'end', query_start for 'start').
    Returns (pred_word [Q] int64, best [Q] float64, scores [Q,L] float64, vecs [Q,L,d] float32).
    """
    Q, d = query.shape
    lut = dequant_lut()
    vecs = np.zeros((Q, L, d), dtype=np.float32)
    new_idx = np.full((Q, L), -1, dtype=np.int64)
    for c in range(Q):
        for slot in range(L):
            i = slot if direction == "end" else (L - 1 - slot)
            w = int(word[c]) + i if direction == "end" else int(word[c]) - i
            ok = (_valid_phrase(index, int(word[c]), w, int(doc[c]), L) if direction == "end"
                  else _valid_phrase(index, w, int(word[c]), int(doc[c]), L))
            new_idx[c, slot] = w if ok else -1
            if branch == "ram":
                gid = int(ids[c]) + i if direction == "end" else int(ids[c]) - i
                if 0 <= gid < index.ntotal:
                    vecs[c, slot] = lut[index.xb[gid].astype(np.int16) + 128]
            else:
                rows = index.docs[int(doc[c])].start
                if direction == "end":
                    if w < rows.shape[0]:
                        vecs[c, slot] = lut[rows[w].astype(np.int16) + 128]
                else:
                    # groups_end: range(max(0, e-L+1), e+1), right-aligned fill (:268-272, 358-361)
                    if w >= 0:
                        vecs[c, slot] = lut[rows[w].astype(np.int16) + 128]
    mask = NEG_MASK * (new_idx < 0)
    dots = np.einsum("qd,qld->ql", query.astype(np.float64), vecs.astype(np.float64))
    dots32 = dots.astype(np.float32)                 # torch fp32 result (:342, :367)
    # index.py:343,368: `np.expand_dims(start_scores, 1) + new_end_scores + end_mask` -- numpy adds left to right: the two
    # fp32 arrays first (an fp32 sum: slots whose dots differ by less than an ulp of the SUM tie, and np.argmax then takes
    # the first of them), the float64 mask after that
    scores = (first_scores.astype(np.float32)[:, None] + dots32).astype(np.float64) + mask
    am = np.argmax(scores, 1)
    pred = new_idx[np.arange(Q), am]
    best = scores[np.arange(Q), am]
    return pred, best, scores, vecs, am


# --------------------------------------------------------------------------
# index.py:167-176
# --------------------------------------------------------------------------
def adjust(each: dict, delimiter: str = " [PAR] ") -> dict:
    ctx = each["context"]
    lo = ctx.rfind(delimiter, 0, each["start_pos"])
    lo = 0 if lo == -1 else lo + len(delimiter)
    hi = ctx.find(delimiter, each["end_pos"])
    hi = len(ctx) if hi == -1 else hi
    each["context"] = ctx[lo:hi]
    each["start_pos"] -= lo
    each["end_pos"] -= lo
    return each


# --------------------------------------------------------------------------
# index.py:178-187: [(X.text, X[0].idx) for X in self.sentencizer(context).sents] -- spaCy 2.3's English tokenizer and
# rule-based `sentencizer`, absent here and restated from memory in oracle/spacy_sentencizer.py (parity unpinned)
# --------------------------------------------------------------------------
def rule_sentences(text: str):
    from .spacy_sentencizer import sentences
    return sentences(text)


def adjust_sent(each: dict) -> dict:
    sents = rule_sentences(each["context"])
    starts = np.array([s for _, s in sents])
    first = int((starts <= each["start_pos"]).sum()) - 1
    last = int((starts <= each["end_pos"] - 1).sum()) - 1
    lo, hi = sorted({first, last})[0], sorted({first, last})[-1]
    each["context"] = " ".join(sents[i][0] for i in range(lo, hi + 1))
    each["start_pos"] -= sents[lo][1]
    each["end_pos"] -= sents[lo][1]
    return each


# --------------------------------------------------------------------------
# index.py:220-422
# --------------------------------------------------------------------------
def search_phrase(index: OracleIndex, query: np.ndarray, sdoc, sword, sI, edoc, eword, eI, sD, eD,
                  top_k: int = 10, max_answer_length: int = 10, return_idxs: bool = False,
                  return_sent: bool = False, branch: str = "ram") -> List[List[dict]]:
    B = query.shape[0]
    L = max_answer_length
    q = np.repeat(query, top_k, axis=0)                       # [B*k, 2d]   (:225)
    qs, qe = np.split(q, 2, axis=1)
    sdoc, sword, edoc, eword = (np.reshape(a, -1) for a in (sdoc, sword, edoc, eword))
    sI, eI = np.reshape(sI, -1), np.reshape(eI, -1)
    sD, eD = np.reshape(sD, -1), np.reshape(eD, -1)

    pred_end, best1, _, end_vecs, am1 = window_rescore(index, qe, sdoc, sword, sI, sD, L, "end", branch)
    pred_start, best2, _, start_vecs, am2 = window_rescore(index, qs, edoc, eword, eI, eD, L, "start", branch)

    n = sdoc.shape[0]
    doc_i = np.stack([sdoc, edoc], 1).reshape(-1)             # interleave (:375-378)
    start_i = np.stack([sword, pred_start], 1).reshape(-1)
    end_i = np.stack([pred_end, eword], 1).reshape(-1)
    score_i = np.stack([best1, best2], 1).reshape(-1)
    if return_idxs:                                           # (:381-389), R = I for a flat index
        # Reference quirk, replicated: the candidate's own vector comes from groups_start/groups_end, which
        # hold *reconstructed floats* in the RAM branch (:282-300) but *raw int8 rows* in the HDF5 branch
        # (:263-272) -- only the window winner (pred_*_vecs) went through dequant in both.
        if branch == "ram":
            own_s, own_e = end_vecs[:, 0, :], start_vecs[:, -1, :]
        else:
            own_s = np.stack([index.docs[int(d_)].start[int(w_)] for d_, w_ in zip(sdoc, sword)]).astype(np.float32)
            own_e = np.stack([index.docs[int(d_)].start[int(w_)] for d_, w_ in zip(edoc, eword)]).astype(np.float32)
        sv = np.stack([own_s, start_vecs[np.arange(n), am2]], 1).reshape(-1, end_vecs.shape[-1])
        ev = np.stack([end_vecs[np.arange(n), am1], own_e], 1).reshape(-1, end_vecs.shape[-1])

    out = []
    for g, (d_, s_, e_, sc) in enumerate(zip(doc_i.tolist(), start_i.tolist(), end_i.tolist(), score_i.tolist())):
        if d_ < 0:
            out.append({"score": DUMMY_SCORE, "context": "dummy", "start_pos": 0, "end_pos": 0, "title": [""]})
            continue
        m = index.docs[d_]
        sp = int(m.word2char_start[m.f2o_start[s_]])
        if len(m.word2char_end) > 0 and e_ >= 0:
            ep = int(m.word2char_end[m.f2o_start[e_]])
        else:
            ep = sp + 1
        out.append({"context": m.context, "title": [m.title], "doc_idx": d_, "start_pos": sp, "end_pos": ep,
                    "start_idx": s_, "end_idx": e_, "score": sc,
                    "start_vec": sv[g] if return_idxs else None,
                    "end_vec": ev[g] if return_idxs else None})
    for each in out:
        each["answer"] = each["context"][each["start_pos"]:each["end_pos"]]
    out = [adjust(each) for each in out]
    if return_sent:
        out = [adjust_sent(each) for each in out]

    grouped: List[List[dict]] = [[] for _ in range(B)]
    per_q = 2 * top_k
    for g, each in enumerate(out):
        grouped[g // per_q].append(each)
    for i in range(B):
        grouped[i] = sorted(grouped[i], key=lambda r: -r["score"])      # stable, like the reference
        grouped[i] = [r for r in grouped[i] if r["score"] > DROP_BELOW]
    return grouped


# --------------------------------------------------------------------------
# eval_utils.py:9-24 (needed by opt4) and index.py:424-448
# --------------------------------------------------------------------------
def normalize_answer(s: str) -> str:
    s = s.lower()
    s = "".join(ch for ch in s if ch not in set(string.punctuation))
    s = re.sub(r"\b(a|an|the)\b", " ", s)
    return " ".join(s.split())


def aggregate_results(results: List[dict], agg_strat: str = "opt1") -> List[dict]:
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
            r["score"] = DUMMY_SCORE
            if agg_strat == "opt4" and r["title"][0] not in results[first[key]]["title"]:
                results[first[key]]["title"] += r["title"]
    results = sorted(results, key=lambda r: -r["score"])
    return [r for r in results if r["score"] > DROP_BELOW]


# --------------------------------------------------------------------------
# index.py:450-482
# --------------------------------------------------------------------------
def search(index: OracleIndex, query: np.ndarray, q_texts: Optional[Sequence[str]] = None, top_k: int = 10,
           aggregate: bool = False, return_idxs: bool = False, max_answer_length: int = 10,
           agg_strat: str = "opt1", return_sent: bool = False, branch: str = "ram") -> List[List[dict]]:
    dense = search_dense(index, query, top_k)
    outs = search_phrase(index, query, *dense, top_k=top_k, max_answer_length=max_answer_length,
                         return_idxs=return_idxs, return_sent=return_sent, branch=branch)
    if aggregate:
        outs = [aggregate_results(r, agg_strat) for r in outs]
    return outs
