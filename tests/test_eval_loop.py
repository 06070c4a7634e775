"""The evaluation loop (tests/_eval_loop.py: test infrastructure since round 6 -- the reference's own eval_phrase_retrieval.py runs unmodified over the product, tests/test_reference_callers.py) against the outputs of the reference's own
eval_phrase_retrieval.py ``evaluate`` / ``evaluate_results`` (run unmodified over the reference's MIPS by
oracle/make_golden_eval.py): EM / F1 at 1 and at k and the per-question prediction records of the ``.pred`` file."""
import json
import os

import numpy as np
import pytest

from tests._golden import GOLD, load_toy_docs

CASES = json.load(open(os.path.join(GOLD, "eval_cases.json")))
_Z = np.load(os.path.join(GOLD, "eval_queries.npz"))
TABLE = {str(t): v for t, v in zip(_Z["texts"].tolist(), _Z["vecs"])}


def test_metric_functions_and_question_loading():
    """CPU: the restated metric functions on hand-checked values, and load_qa_pairs on the golden QA file"""
    from tests._eval_loop import exact_match_score, f1_score, load_qa_pairs, normalize_answer, regex_match_score
    assert normalize_answer("The  Quick, brown fox!") == "quick brown fox"
    assert exact_match_score("the Eiffel Tower.", "Eiffel tower") and not exact_match_score("Eiffel", "Eiffel tower")
    assert abs(f1_score("big red dog", "red dog") - 0.8) < 1e-12 and f1_score("yes", "no") == 0.0 and f1_score("cat", "dog") == 0.0
    assert regex_match_score("1999", r"19\d\d") and not regex_match_score("x", "(")
    qids, questions, answers, titles = load_qa_pairs(os.path.join(GOLD, "eval_qa.json"))
    assert len(qids) == 7 and all(not q.endswith("?") for q in questions) and set(questions) == set(TABLE)
    rec = CASES[0]["pred"]
    assert [rec[q]["question"] for q in qids] == questions and [rec[q]["answer"] for q in qids] == answers


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(len(CASES)))
def test_eval_loop_matches_the_reference(ci):
    from densephrases_amd import DocMeta, DocStore, MIPS
    from tests._eval_loop import evaluate
    c = CASES[ci]
    store = DocStore([DocMeta(m.doc_idx, m.title, m.context, m.f2o_start, m.word2char_start, m.word2char_end, m.start)
                      for m in load_toy_docs()])
    mips = MIPS.from_store(store)
    enc = lambda qs: [(TABLE[q][None, :768].tolist(), TABLE[q][None, 768:].tolist(), q.split()) for q in qs]      # noqa: E731
    em1, f11, emk, f1k, pred = evaluate(mips, enc, os.path.join(GOLD, "eval_qa.json"), top_k=c["top_k"],
                                        eval_batch_size=c["eval_batch_size"], aggregate=c["aggregate"], agg_strat=c["agg_strat"])
    np.testing.assert_allclose([em1, f11, emk, f1k], c["metrics"], rtol=0, atol=1e-9)
    want = c["pred"]
    assert list(pred.keys()) == list(want.keys())
    k = c["top_k"]
    for qid, w in want.items():
        g = pred[qid]
        for key in ("question", "answer", "prediction", "title", "evidence", "em_top1", f"em_top{k}", "rd_topk"):
            assert g[key] == w[key], (qid, key, g[key], w[key])
        assert [list(p) for p in g["se_pos"]] == [list(p) for p in w["se_pos"]]
        np.testing.assert_allclose(g["score"], w["score"], rtol=1e-6, atol=1e-4)
        np.testing.assert_allclose([g["f1_top1"], g[f"f1_top{k}"]], [w["f1_top1"], w[f"f1_top{k}"]], atol=1e-12)
