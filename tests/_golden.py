"""Helpers to rebuild the committed golden dump (tests/golden/toy_dump.npz) as oracle DocMeta objects."""
import json
import os

import numpy as np

from oracle.mips_oracle import DocMeta, build_index_from_docs

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_toy_docs():
    z = np.load(os.path.join(GOLD, "toy_dump.npz"))
    docs = []
    for i, di in enumerate(z["doc_ids"].tolist()):
        docs.append(DocMeta(doc_idx=int(di), title=str(z["titles"][i]), context=str(z["contexts"][i]),
                            f2o_start=z[f"f2o_{di}"], word2char_start=z[f"w2cs_{di}"],
                            word2char_end=z[f"w2ce_{di}"], start=z[f"start_{di}"]))
    return docs


def load_toy_index():
    return build_index_from_docs(load_toy_docs())


def load_cases():
    with open(os.path.join(GOLD, "toy_cases.json")) as f:
        cases = json.load(f)
    z = np.load(os.path.join(GOLD, "toy_vecs.npz"))
    for c in cases:
        c["query_arr"] = z[f"query_{c['query']}"]
    return cases, z["vecs"]


def compare_results(got, want, vecs, score_rtol=1e-6, score_atol=1e-4, case=None):
    """got: List[List[dict]] from an implementation; want: the reference's output (json-ified).  ``case``: the golden
    case; its near-tie queries (make_golden.py) carry components of ~1e3, so one of their window dots is a sum with heavy
    cancellation that torch's fp32 reduction (the reference) and an exactly rounded dot evaluate 2e-6 apart -- ids,
    positions and the arg-max slots must still agree exactly, the score tolerance is widened to 2e-5."""
    if case is not None and case.get("near_tie") and score_rtol > 0:
        score_rtol = max(score_rtol, 2e-5)
    assert len(got) == len(want)
    for qi, (g, w) in enumerate(zip(got, want)):
        assert len(g) == len(w), f"query {qi}: {len(g)} results vs reference {len(w)}"
        for ri, (a, b) in enumerate(zip(g, w)):
            for key in ("context", "title", "doc_idx", "start_pos", "end_pos", "start_idx", "end_idx", "answer"):
                assert a[key] == b[key], f"query {qi} result {ri} field {key}: {a[key]!r} != {b[key]!r}"
            assert np.isclose(a["score"], b["score"], rtol=score_rtol, atol=score_atol), (qi, ri, a["score"], b["score"])
            for key in ("start_vec", "end_vec"):
                if b.get(key) is None:
                    assert a.get(key) is None
                else:
                    np.testing.assert_allclose(np.asarray(a[key], np.float32), vecs[b[key]], rtol=1e-6, atol=1e-6)


def load_custom():
    """The reference's own example (examples/create-custom-index: article text + questions; oracle/make_golden_custom.py) ->
    (docs [DocMeta, seeded int8 rows], query table, questions, golden cases)"""
    from oracle.make_golden_custom import docs_from_text, query_table
    with open(os.path.join(GOLD, "custom_dump.json")) as f:
        dump = json.load(f)
    with open(os.path.join(GOLD, "custom_cases.json")) as f:
        cases = json.load(f)
    docs = docs_from_text([tuple(t) for t in dump["docs"]], seed=dump["seed"])
    table, where = query_table(docs, dump["questions"], seed=dump["seed"])
    assert {k: list(map(int, v)) for k, v in where.items()} == cases["answer_tokens"]
    return docs, table, dump["questions"], cases
