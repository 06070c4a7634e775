#!/usr/bin/env python3
"""Goldens for the PQ branch: the REFERENCE's own MIPS (densephrases/index.py, loaded unmodified through oracle/refshim)
over a real ``index.faiss`` + ``merged.invdata`` pair in the FAISS 1.6 format (densephrases_amd/faiss_io.py) holding an
IndexPreTransform(OPQ 768x768) -> IndexIVFPQ(nlist 4, M 96, 8 bits, inner product, by_residual) over the toy dump (ids =
row numbers, one idx2id group at offset 0: the reference's get_idxs clips every id >= ntotal, index.py:128-133, so an
index with id offsets only works for it when the sub-dumps are full; ids beyond 2^32 are covered by tests/test_pq.py).
The refshim ``faiss`` answers ``search`` / ``reconstruct`` with oracle/ivfpq_oracle.py (FAISS itself is absent: the dense
half is the restatement, everything above it -- get_idxs with offsets, the reconstruct loops, ``@ R``, windows, masks,
the doubly rotated ``pred_*_vecs`` of return_idxs, dict assembly -- is reference code).

    python -m oracle.make_golden_pq          (build container; needs /root/reference)

Writes tests/golden/pq_index.npz (the index pieces: tests rebuild identical files from them with faiss_io.write_index),
pq_cases.json, pq_vecs.npz."""
from __future__ import annotations

import json
import math
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from densephrases_amd import faiss_io                    # noqa: E402
from oracle import ivfpq_oracle as P                      # noqa: E402
from oracle import refshim                                # noqa: E402
from oracle.make_golden import jsonable, write_reference_layout      # noqa: E402
from oracle.synth_dump import make_queries                # noqa: E402

MAX_IDX = int(1e9)
INDEX_NAME = "toy_OPQ96_PQ"


def householder_rotation(v: np.ndarray, perm: np.ndarray) -> np.ndarray:
    """An orthogonal 768 x 768 matrix from 768 + 768 numbers, bit-identical on every machine: P (I - 2 v v^T / <v,v>),
    every operation element-wise in fp32 (no BLAS summation order involved)."""
    v = np.asarray(v, np.float32)
    vv = np.float32(math.fsum(float(x) * float(x) for x in v))
    H = np.eye(len(v), dtype=np.float32) - (np.float32(2.0) / vv) * np.outer(v, v).astype(np.float32)
    return np.ascontiguousarray(H[np.asarray(perm)])


def build_index(pieces) -> faiss_io.PreTransformIndex:
    """the index object from the committed pieces (also used by the tests)"""
    A = householder_rotation(pieces["v"], pieces["perm"])
    nlist = int(pieces["centroids"].shape[0])
    codes, ids = pieces["codes"], pieces["ids"]
    sizes = pieces["list_sizes"].astype(np.int64)
    off = np.concatenate([[0], np.cumsum(sizes)])
    ivf = faiss_io.IVFPQIndex(768, nlist, int(pieces["pq_centroids"].shape[0]), 8, pieces["centroids"].astype(np.float32),
                              pieces["pq_centroids"].astype(np.float32), [codes[off[l]:off[l + 1]] for l in range(nlist)],
                              [ids[off[l]:off[l + 1]] for l in range(nlist)], True, 0, 1, 2)
    return faiss_io.PreTransformIndex([faiss_io.LinearTransform(A)], ivf, 768, True)


def split_ids(n_rows: int, cut: int) -> np.ndarray:
    """FAISS ids of the stored rows: sub-dump 0 holds rows [0, cut) under ids 0.., sub-dump 1 the rest under 1e9.."""
    ids = np.arange(n_rows, dtype=np.int64)
    ids[cut:] = MAX_IDX + np.arange(n_rows - cut)
    return ids


def main():
    from tests._golden import load_toy_docs
    ref = refshim.install()
    docs = load_toy_docs()
    rng = np.random.default_rng(77)
    cases, vec_store, query_store = [], [], []
    with tempfile.TemporaryDirectory() as tmp:
        dump_dir, idx = write_reference_layout(tmp, docs, INDEX_NAME)
        xb = idx.xb.astype(np.float32) / np.float32(20.0) + np.float32(-2.0)
        n = xb.shape[0]
        # the cut between the two sub-dumps falls on a document boundary, like two dumps built apart
        order = [m for m in sorted(docs, key=lambda m: str(m.doc_idx)) if m.start.shape[0] > 0]
        cut = int(sum(m.start.shape[0] for m in order[:len(order) // 2]))
        ids = np.arange(n, dtype=np.int64)
        # train: coarse + PQ codebooks by k-means on the rotated vectors, rotation = Householder x permutation
        v = rng.normal(0, 1, 768).astype(np.float32)
        perm = rng.permutation(768).astype(np.int64)
        A = householder_rotation(v, perm)
        trained = P.train(xb, nlist=4, M=96, seed=5, rotate=False)
        chain = [faiss_io.LinearTransform(A)]
        xp = P.apply_chain(chain, xb)
        t2 = P.train(xp, nlist=4, M=96, seed=5, rotate=False)               # codebooks fitted to the ROTATED vectors
        index = faiss_io.PreTransformIndex(chain, t2.index, 768, True)
        P.add_with_ids(index, xb, ids)
        del trained
        ivf = index.index
        pieces = {"v": v, "perm": perm, "centroids": ivf.centroids, "pq_centroids": ivf.pq_centroids,
                  "codes": np.concatenate(ivf.list_codes), "ids": np.concatenate(ivf.list_ids),
                  "list_sizes": np.asarray([len(i) for i in ivf.list_ids], np.int64), "cut": np.asarray([cut], np.int64)}
        rebuilt = build_index(pieces)
        assert (rebuilt.chain[0].A == A).all()
        idx_dir = os.path.join(dump_dir, "start", INDEX_NAME)
        faiss_io.write_index(rebuilt, os.path.join(idx_dir, "index.faiss"), ondisk=True)
        mips = ref.MIPS(phrase_dump_dir=os.path.join(dump_dir, "phrase"), index_path=os.path.join(idx_dir, "index.faiss"),
                        idx2id_path=os.path.join(idx_dir, "idx2id.hdf5"), cuda=False)
        assert mips.index.ntotal == n and mips.max_idx == 1e9 and mips.doc_groups is not None
        for (B, k, L, agg, strat, ridx, sent) in [
            (4, 5, 10, False, "opt1", False, False),
            (4, 5, 10, True, "opt1", False, False),
            (3, 10, 3, True, "opt3", False, False),
            (2, 3, 10, False, "opt1", True, False),
            (3, 8, 10, True, "opt2", False, True),
            (1, 20, 10, False, "opt1", False, False),
        ]:
            q = make_queries(rng, idx.xb, B)
            query_store.append(q.astype(np.float32))
            dense = mips.search_dense(q, q_texts=None, top_k=k)
            res = mips.search(q.astype(np.float64), q_texts=[f"q{i}" for i in range(B)], top_k=k, aggregate=agg,
                              return_idxs=ridx, max_answer_length=L, agg_strat=strat, return_sent=sent)
            cases.append({"branch": "pq", "B": B, "top_k": k, "L": L, "aggregate": agg, "agg_strat": strat, "return_idxs": ridx,
                          "return_sent": sent, "query": len(query_store) - 1,
                          "dense": [np.asarray(a).tolist() for a in dense], "results": jsonable(res, vec_store)})
    gold = os.path.join(ROOT, "tests", "golden")
    np.savez_compressed(os.path.join(gold, "pq_index.npz"), **pieces)
    with open(os.path.join(gold, "pq_cases.json"), "w") as f:
        json.dump(cases, f)
    np.savez_compressed(os.path.join(gold, "pq_vecs.npz"), vecs=np.stack(vec_store) if vec_store else np.zeros((0, 768), np.float32),
                        **{f"query_{i}": q for i, q in enumerate(query_store)})
    print(f"wrote {len(cases)} PQ cases, {len(vec_store)} vectors, index pieces {os.path.getsize(os.path.join(gold, 'pq_index.npz'))} bytes")


if __name__ == "__main__":
    main()
