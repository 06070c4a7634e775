#!/usr/bin/env python3
"""Generate tests/golden/custom_*.json from the reference's OWN example inputs: the text of
/root/reference/examples/create-custom-index/articles.json (4 Wikipedia articles; two paragraphs of each are kept: the first and the
one holding the example's answer) and its questions.json (3 questions with answers), run through the reference's own ``evaluate``
(eval_phrase_retrieval.py:49-205, unmodified) over the reference's own ``MIPS`` (index.py through oracle/refshim).

The example ships text only -- no vectors (its README builds them with the trained encoder, which does not exist offline) -- so the
dump's int8 rows are synthetic (seeded, re-drawn by tests/_golden.load_custom_docs: the golden stores text and seed, not 1.2 MB of
noise), tokens are the whitespace tokens of the text with their character offsets, and the query vector of a question is the row of
its answer's first / last token plus noise: the path from vectors back to the answer STRING (idx2id, f2o, word2char, paragraph
cropping, EM scoring) is the reference's, on the reference's text, and the expected top-1 is the example's own answer.

Run from the repo root in the build container (needs /root/reference):   python -m oracle.make_golden_custom
"""
from __future__ import annotations

import argparse
import json
import os
import re
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import mips_oracle as O                          # noqa: E402
from oracle.make_golden import write_reference_layout        # noqa: E402
from oracle.refshim import callers, REFERENCE_ROOT           # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
EXAMPLE = os.path.join(REFERENCE_ROOT, "examples", "create-custom-index")
SEED = 4242


def docs_from_text(texts, seed=SEED):
    """[(doc_idx, title, context)] -> DocMeta list: whitespace tokens with character offsets, every token kept, seeded int8 rows"""
    rng = np.random.default_rng(seed)
    docs = []
    for doc_idx, title, context in texts:
        spans = [(m.start(), m.end()) for m in re.finditer(r"\S+", context) if m.group() != "[PAR]"]
        n = len(spans)
        start = O.float_to_int8(rng.normal(0.0, 0.6, size=(n, 768)).astype(np.float32))
        docs.append(O.DocMeta(doc_idx=int(doc_idx), title=title, context=context, f2o_start=np.arange(n, dtype=np.int64),
                              word2char_start=np.asarray([s for s, _ in spans], np.int32),
                              word2char_end=np.asarray([e for _, e in spans], np.int32), start=start))
    return docs


def answer_span(doc, answer):
    """(first token, last token) of the first occurrence of `answer` as whole tokens (trailing punctuation allowed on the last)"""
    words = answer.split()
    toks = [doc.context[s:e] for s, e in zip(doc.word2char_start, doc.word2char_end)]
    for i in range(len(toks) - len(words) + 1):
        if toks[i:i + len(words) - 1] == words[:-1] and toks[i + len(words) - 1].rstrip(".,;:)") == words[-1]:
            return i, i + len(words) - 1
    raise ValueError(answer)


def query_table(docs, questions, seed=SEED):
    """question text (as load_qa_pairs hands it on) -> (start vec, end vec): the answer's first / last token row + noise"""
    rng = np.random.default_rng(seed + 1)
    table, where = {}, {}
    for q in questions:
        for d in docs:
            try:
                s, e = answer_span(d, q["answers"][0])
            except ValueError:
                continue
            text = q["question"][:-1] if q["question"].endswith("?") else q["question"]
            x = O.int8_to_float(d.start)
            table[text] = ((x[s] + rng.normal(0, 0.1, 768)).astype(np.float32), (x[e] + rng.normal(0, 0.1, 768)).astype(np.float32))
            where[q["id"]] = (d.doc_idx, s, e)
            break
        else:
            raise ValueError(q)
    return table, where


def eval_args(test_path, load_dir, top_k=5):
    return argparse.Namespace(test_path=test_path, do_lower_case=False, draft=False, truecase=False, cuda=False, eval_batch_size=2,
                              nprobe=256, top_k=top_k, max_answer_length=10, aggregate=True, agg_strat="opt1", return_sent=False,
                              is_kilt=False, candidate_path=None, regex=False, load_dir=load_dir, save_pred=True, eval_psg=False,
                              verbose_logging=False, max_query_length=64)


def main():
    arts = json.load(open(os.path.join(EXAMPLE, "articles.json")))["data"]
    questions = json.load(open(os.path.join(EXAMPLE, "questions.json")))["data"]
    texts = []
    for i, a in enumerate(arts):
        keep = [0]
        for q in questions:
            for pi, p in enumerate(a["paragraphs"]):
                if re.search(r"(^|\s)" + re.escape(q["answers"][0]) + r"([\s.,;:)]|$)", p["context"]) and pi not in keep:
                    keep.append(pi)
                    break
        if len(keep) == 1:
            keep.append(1)
        texts.append((100 + i, a["title"], " [PAR] ".join(a["paragraphs"][pi]["context"].strip() for pi in sorted(keep[:2]))))
    docs = docs_from_text(texts)
    table, where = query_table(docs, questions)
    os.makedirs(GOLD, exist_ok=True)
    with open(os.path.join(GOLD, "custom_dump.json"), "w") as f:
        json.dump({"source": "examples/create-custom-index/articles.json + questions.json of the reference (text only; vectors: seed)",
                   "seed": SEED, "docs": [[d, t, c] for d, t, c in texts], "questions": questions}, f)
    ref_index, ou, model, ev = callers.install_callers(None, table)
    with tempfile.TemporaryDirectory() as tmp:
        dump_dir, idx = write_reference_layout(os.path.join(tmp, "hdf5"), docs, "custom_flat_none")
        mips = ref_index.MIPS(phrase_dump_dir=os.path.join(dump_dir, "phrase"),
                              index_path=os.path.join(dump_dir, "start", "custom_flat_none", "index.faiss"),
                              idx2id_path=os.path.join(dump_dir, "start", "custom_flat_none", "idx2id.hdf5"), cuda=False)
        qa = os.path.join(tmp, "questions.json")
        json.dump({"data": questions}, open(qa, "w"))
        args = eval_args(qa, os.path.join(tmp, "run"))
        em1, f11, emk, f1k = ev.evaluate(args, mips=mips, query_encoder=object(), tokenizer=None)
        pred_file = [p for p in os.listdir(os.path.join(args.load_dir, "pred")) if p.endswith(".pred")][0]
        pred = json.load(open(os.path.join(args.load_dir, "pred", pred_file)))
    assert em1 == 100.0, (em1, pred)                              # the example's own answers come back first
    with open(os.path.join(GOLD, "custom_cases.json"), "w") as f:
        json.dump({"top_k": args.top_k, "metrics": [em1, f11, emk, f1k], "pred_file": pred_file, "pred": pred,
                   "answer_tokens": {k: list(map(int, v)) for k, v in where.items()}}, f)
    print("rows", sum(d.start.shape[0] for d in docs), "metrics", [em1, f11, emk, f1k], {k: pred[k]["prediction"][0] for k in pred})


if __name__ == "__main__":
    main()
