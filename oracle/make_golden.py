#!/usr/bin/env python3
"""Generate tests/golden/* by running the REFERENCE's own MIPS class (loaded unmodified from
/root/reference/densephrases/index.py through oracle/refshim) on small synthetic dumps.

Run from the repo root in the build container (needs /root/reference):

    python -m oracle.make_golden

Outputs (committed):
    tests/golden/toy_dump.npz            the dump (so tests rebuild identical inputs without /root/reference)
    tests/golden/toy_cases.json          per-case inputs + the reference's outputs
    tests/golden/toy_vecs.npz            start_vec / end_vec arrays of the return_idxs case

Both metadata branches of index.py are exercised: the HDF5 branch (index path without 'PQ',
index.py:246-273) and the RAM branch ('PQ' in the index path + meta_compressed.pkl, index.py:276-302).
"""
from __future__ import annotations

import json
import os
import pickle
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import refshim                      # noqa: E402
from oracle.mips_oracle import build_index_from_docs  # noqa: E402
from oracle.synth_dump import make_dump, make_queries  # noqa: E402


def write_reference_layout(tmp, docs, index_name):
    import zlib
    dump_dir = os.path.join(tmp, "dump")
    os.makedirs(os.path.join(dump_dir, "phrase"), exist_ok=True)
    os.makedirs(os.path.join(dump_dir, "start", index_name), exist_ok=True)
    groups = {}
    for m in docs:
        groups[str(m.doc_idx)] = {
            "attrs": {"context": m.context, "title": m.title, "offset": -2.0, "scale": 20.0},
            "data": {"start": m.start, "f2o_start": m.f2o_start,
                     "word2char_start": m.word2char_start, "word2char_end": m.word2char_end},
        }
    refshim.write_fake_h5(os.path.join(dump_dir, "phrase", "0-1.hdf5"), groups)
    idx = build_index_from_docs(docs)
    with open(os.path.join(dump_dir, "start", index_name, "index.faiss"), "wb") as f:
        pickle.dump({"xb": idx.xb}, f)
    refshim.write_fake_h5(os.path.join(dump_dir, "start", index_name, "idx2id.hdf5"),
                          {"0": {"attrs": {"offset": 0}, "data": {"doc": idx.row2doc, "word": idx.row2word}}})
    meta = {}
    for m in docs:   # scripts/preprocess/compress_metadata.py:45-53 layout
        meta[str(m.doc_idx)] = {
            "word2char_start": zlib.compress(m.word2char_start.tobytes()),
            "word2char_end": zlib.compress(m.word2char_end.tobytes()),
            "f2o_start": zlib.compress(m.f2o_start.tobytes()),
            "context": zlib.compress(m.context.encode("utf-8")),
            "title": m.title,
            "dtypes": {"word2char_start": m.word2char_start.dtype, "word2char_end": m.word2char_end.dtype,
                       "f2o_start": m.f2o_start.dtype},
        }
    with open(os.path.join(dump_dir, "meta_compressed.pkl"), "wb") as f:
        pickle.dump(meta, f)
    return dump_dir, idx


def jsonable(results, vec_store):
    out = []
    for per_q in results:
        rows = []
        for r in per_q:
            row = {}
            for k, v in r.items():
                if k in ("start_vec", "end_vec"):
                    if v is None:
                        row[k] = None
                    else:
                        row[k] = len(vec_store)
                        vec_store.append(np.asarray(v, np.float32))
                elif isinstance(v, (np.integer,)):
                    row[k] = int(v)
                elif isinstance(v, (np.floating,)):
                    row[k] = float(v)
                else:
                    row[k] = v
            rows.append(row)
        out.append(rows)
    return out


def main():
    ref = refshim.install()
    docs = make_dump(seed=42, n_docs=6, d=768)
    # make two rows exact duplicates (ties) and one doc tiny (window runs off the doc end)
    docs[1].start[3] = docs[0].start[2]
    docs.append(make_dump(seed=7, n_docs=1, d=768, doc_ids=[905], n_par=1, words_per_par=(2, 3), keep_prob=1.0)[0])
    rng = np.random.default_rng(123)
    cases = []
    vec_store = []
    query_store = []
    made = {}
    with tempfile.TemporaryDirectory() as tmp:
        for branch, index_name in (("hdf5", "toy_flat_none"), ("ram", "toy_flat_PQ96")):
            dump_dir, idx = write_reference_layout(os.path.join(tmp, branch), docs, index_name)
            mips = ref.MIPS(phrase_dump_dir=os.path.join(dump_dir, "phrase"),
                            index_path=os.path.join(dump_dir, "start", index_name, "index.faiss"),
                            idx2id_path=os.path.join(dump_dir, "start", index_name, "idx2id.hdf5"),
                            cuda=False)
            made[branch] = (mips, idx)
            for (B, k, L, agg, strat, ridx, sent) in [
                (4, 5, 10, False, "opt1", False, False),
                (4, 5, 10, True, "opt1", False, False),
                (3, 10, 10, True, "opt2", False, False),
                (3, 10, 3, True, "opt3", False, False),
                (3, 4, 10, True, "opt4", False, False),
                (1, 10, 10, False, "opt1", False, False),
                (2, 3, 10, False, "opt1", True, False),
                (2, 4, 10, True, "opt2", False, True),
            ]:
                q = make_queries(rng, idx.xb, B)
                query_store.append(q.astype(np.float32))
                dense = mips.search_dense(q, q_texts=None, top_k=k)
                res = mips.search(q.astype(np.float64), q_texts=[f"q{i}" for i in range(B)], top_k=k,
                                  aggregate=agg, return_idxs=ridx, max_answer_length=L, agg_strat=strat,
                                  return_sent=sent)
                cases.append({
                    "branch": branch, "B": B, "top_k": k, "L": L, "aggregate": agg, "agg_strat": strat,
                    "return_idxs": ridx, "return_sent": sent,
                    "query": len(query_store) - 1,
                    "dense": [np.asarray(a).tolist() for a in dense],
                    "results": jsonable(res, vec_store),
                })
        # Appended after the 16 cases above (their indices and queries stay what they were): WINDOW NEAR-TIES.  The
        # reference adds the first-stage score and the window dot as fp32 numbers (index.py:343,368), so slots whose dots
        # differ by less than an ulp of the SUM collapse to a tie and np.argmax takes the first of them.  To make that
        # happen for sure and independently of any summation order, the window half of the query is a single 1.0 (the dot
        # IS one de-quantised component: exact in every implementation) and the first-stage half is a stored row scaled
        # until its scores are ~5e7 (ulp 4, components range over [-8.4, 4.35]).
        rng2 = np.random.default_rng(2027)
        for branch in ("hdf5", "ram"):
            mips, idx = made[branch]
            B, k, L = 3, 6, 10
            rows = rng2.choice(idx.xb.shape[0], B, replace=False)
            q = np.zeros((B, 1536), dtype=np.float32)
            for b, r in enumerate(rows):
                x = idx.xb[r].astype(np.float32) / np.float32(20.0) + np.float32(-2.0)
                q[b, :768] = x * np.float32(5.0e7 / float(x @ x))             # <q_start, x_r> ~ 5e7
                q[b, 768 + int(rng2.integers(0, 768))] = 1.0                  # <q_end, v> = v[j]
            q[B - 1, :768], q[B - 1, 768:] = q[B - 1, 768:].copy(), q[B - 1, :768].copy()    # and once the other way round
            query_store.append(q)
            dense = mips.search_dense(q, q_texts=None, top_k=k)
            res = mips.search(q.astype(np.float64), q_texts=[f"q{i}" for i in range(B)], top_k=k, aggregate=False,
                              return_idxs=False, max_answer_length=L, agg_strat="opt1", return_sent=False)
            cases.append({"branch": branch, "B": B, "top_k": k, "L": L, "aggregate": False, "agg_strat": "opt1",
                          "return_idxs": False, "return_sent": False, "near_tie": True, "query": len(query_store) - 1,
                          "dense": [np.asarray(a).tolist() for a in dense], "results": jsonable(res, vec_store)})
    gold = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gold, exist_ok=True)
    np.savez_compressed(
        os.path.join(gold, "toy_dump.npz"),
        doc_ids=np.asarray([m.doc_idx for m in docs], np.int64),
        titles=np.asarray([m.title for m in docs]),
        contexts=np.asarray([m.context for m in docs]),
        **{f"start_{m.doc_idx}": m.start for m in docs},
        **{f"f2o_{m.doc_idx}": m.f2o_start for m in docs},
        **{f"w2cs_{m.doc_idx}": m.word2char_start for m in docs},
        **{f"w2ce_{m.doc_idx}": m.word2char_end for m in docs},
    )
    with open(os.path.join(gold, "toy_cases.json"), "w") as f:
        json.dump(cases, f)
    np.savez_compressed(os.path.join(gold, "toy_vecs.npz"), vecs=np.stack(vec_store),
                        **{f"query_{i}": q for i, q in enumerate(query_store)})
    print(f"wrote {len(cases)} cases, {len(vec_store)} vectors")


if __name__ == "__main__":
    main()
