#!/usr/bin/env python3
"""Generate tests/golden/train_query.json by running the REFERENCE's second caller of ``MIPS.search`` -- ``get_top_phrases``
(/root/reference/train_query.py:182-205: top_k = 100, return_idxs = True, the query encoder through open_utils.get_query2vec) and
``annotate_phrase_vecs`` (:208-275: padding to 2 * top_k, zero-masked start / end vectors, phrase- and document-level targets) --
loaded unmodified (oracle/refshim/callers.py) over the reference's own ``MIPS`` (index.py through oracle/refshim) on the toy dump,
with the questions / answers / titles of tests/golden/eval_qa.json and the stored query vectors of eval_queries.npz (the encoder
stand-in of the other caller goldens).

The golden holds, per question: the phrases (doc_idx, start_idx, end_idx, answer, score), the targets and p_targets, and -- instead
of the 2 x 200 x 768 floats -- the CHECK that every start / end vector the reference returned is the fp32 de-quantised row of its
(doc_idx, start_idx / end_idx), which the test repeats over the product.

Run from the repo root in the build container (needs /root/reference):   python -m oracle.make_golden_train_query
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import mips_oracle as O                          # noqa: E402
from oracle.make_golden import write_reference_layout        # noqa: E402
from oracle.refshim import callers                           # noqa: E402
from tests._golden import GOLD, load_toy_docs                # noqa: E402

TOP_K = 100


def query_table():
    z = np.load(os.path.join(GOLD, "eval_queries.npz"))
    return {str(t): (v[:768].astype(np.float32), v[768:].astype(np.float32)) for t, v in zip(z["texts"].tolist(), z["vecs"])}


def train_args(**over):
    a = argparse.Namespace(cuda=False, max_query_length=64, verbose_logging=False, nprobe=256, top_k=TOP_K, max_answer_length=10,
                           aggregate=True, agg_strat="opt2", label_strat="phrase,doc", regex=False, do_lower_case=False, draft=False,
                           truecase=False)
    a.__dict__.update(over)
    return a


def row_vectors(docs):
    """(doc_idx, word) -> fp32 de-quantised start vector of that row (index.py:282-300 reconstructs exactly these)"""
    out = {}
    for m in docs:
        x = O.int8_to_float(m.start)
        for w in range(x.shape[0]):
            out[(int(m.doc_idx), w)] = x[w]
    return out


def run_caller(tq, ou, mips, docs, batch_size=3):
    """the reference's two functions over `mips`, batch by batch -> (records per question, svs, evs per batch)"""
    args = train_args()
    q_ids, questions, answers, titles = ou.load_qa_pairs(os.path.join(GOLD, "eval_qa.json"), args)
    recs, vec_batches = [], []
    for b_ids, b_q, b_a, b_t, outs in tq.get_top_phrases(mips, q_ids, questions, answers, titles, object(), None, batch_size, args):
        groups = [[dict(o) for o in out] for out in outs]
        svs, evs, tgts, p_tgts = tq.annotate_phrase_vecs(mips, b_ids, b_q, b_a, b_t, groups, args)
        vec_batches.append((svs, evs, groups))
        for i, qid in enumerate(b_ids):
            recs.append({"q_id": qid, "n_phrases": len(outs[i]),
                         "phrases": [[int(p["doc_idx"]), int(p["start_idx"]), int(p["end_idx"]), p["answer"], float(p["score"])] for p in outs[i]],
                         "targets": tgts[i], "p_targets": p_tgts[i]})
    return recs, vec_batches


def check_vectors(vec_batches, rows):
    """annotate_phrase_vecs' [B, 2 top_k, 768] arrays: the fp32 row of every real phrase, zeros for the padding"""
    for svs, evs, groups in vec_batches:
        assert svs.shape == evs.shape == (len(groups), 2 * TOP_K, 768)
        for b, g in enumerate(groups):
            assert len(g) == 2 * TOP_K
            for j, p in enumerate(g):
                if int(p["doc_idx"]) < 0:
                    assert not svs[b, j].any() and not evs[b, j].any()
                else:
                    np.testing.assert_array_equal(svs[b, j].astype(np.float32), rows[(int(p["doc_idx"]), int(p["start_idx"]))])
                    np.testing.assert_array_equal(evs[b, j].astype(np.float32), rows[(int(p["doc_idx"]), int(p["end_idx"]))])


def main():
    docs = load_toy_docs()
    ref_index, ou, model, ev = callers.install_callers(None, query_table())
    tq = callers.load_train_query()
    with tempfile.TemporaryDirectory() as tmp:
        dump_dir, idx = write_reference_layout(os.path.join(tmp, "ram"), docs, "toy_flat_PQ96")     # the RAM branch: reconstructed float vectors
        mips = ref_index.MIPS(phrase_dump_dir=os.path.join(dump_dir, "phrase"),
                              index_path=os.path.join(dump_dir, "start", "toy_flat_PQ96", "index.faiss"),
                              idx2id_path=os.path.join(dump_dir, "start", "toy_flat_PQ96", "idx2id.hdf5"), cuda=False)
        recs, vec_batches = run_caller(tq, ou, mips, docs)
    check_vectors(vec_batches, row_vectors(docs))
    with open(os.path.join(GOLD, "train_query.json"), "w") as f:
        json.dump({"top_k": TOP_K, "batch_size": 3, "records": recs}, f)
    print("wrote", len(recs), "questions;", [r["n_phrases"] for r in recs], "phrases;",
          sum(t is not None for r in recs for t in r["targets"]), "phrase targets,",
          sum(t is not None for r in recs for t in r["p_targets"]), "doc targets")


if __name__ == "__main__":
    main()
