#!/usr/bin/env python3
"""Generate tests/golden/eval_*.json by running the REFERENCE's own evaluation loop -- ``evaluate`` and
``evaluate_results`` of /root/reference/eval_phrase_retrieval.py (:49-205), loaded unmodified -- over the reference's own
``MIPS`` (index.py through oracle/refshim) on the toy dump, with the reference's own ``load_qa_pairs``
(open_utils.py:104-160, extracted from the file at run time) and metric functions (densephrases/utils/eval_utils.py,
loaded as is).  Stubbed, and only that: the query encoder (``get_query2vec`` returns a table look-up, as in
oracle/make_golden_model.py) and the modules eval_phrase_retrieval.py imports but this path never calls (kilt, requests,
load_encoder, Options).

Run from the repo root in the build container (needs /root/reference):   python -m oracle.make_golden_eval
"""
from __future__ import annotations

import argparse
import ast
import importlib.util
import json
import logging
import os
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import refshim                                  # noqa: E402
from oracle.make_golden import write_reference_layout       # noqa: E402
from oracle.make_golden_model import TableEncoder           # noqa: E402
from oracle.synth_dump import make_queries                  # noqa: E402
from tests._golden import load_toy_docs                     # noqa: E402

REF = refshim.REFERENCE_ROOT


def _function_from_file(path, name, namespace):
    """compile ONE top-level function of a reference file in `namespace` (the file's other imports are not executed)"""
    src = open(path).read()
    tree = ast.parse(src)
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    code = compile(ast.Module(body=[node], type_ignores=[]), path, "exec")
    exec(code, namespace)
    return namespace[name]


def load_reference_eval(table):
    ref_index = refshim.install()
    pkg = sys.modules["densephrases"]
    pkg.Options = type("Options", (), {})
    pkg.MIPS = ref_index.MIPS
    utils = types.ModuleType("densephrases.utils")
    utils.__path__ = [f"{REF}/densephrases/utils"]
    sys.modules["densephrases.utils"] = utils
    # the metric functions: the reference's file, as is (it only needs ujson, which refshim provides)
    spec = importlib.util.spec_from_file_location("densephrases.utils.eval_utils", f"{REF}/densephrases/utils/eval_utils.py")
    ev = importlib.util.module_from_spec(spec)
    sys.modules["densephrases.utils.eval_utils"] = ev
    spec.loader.exec_module(ev)
    su = types.ModuleType("densephrases.utils.single_utils")
    su.load_encoder = None
    sys.modules["densephrases.utils.single_utils"] = su
    ou = types.ModuleType("densephrases.utils.open_utils")
    ns = {"json": json, "os": os, "np": np, "random": __import__("random"), "logger": logging.getLogger("open_utils"),
          "truecase": None, "TrueCaser": None}
    ou.load_qa_pairs = _function_from_file(f"{REF}/densephrases/utils/open_utils.py", "load_qa_pairs", ns)
    ou.load_phrase_index = None
    ou.get_query2vec = lambda query_encoder, tokenizer, args, batch_size=64: TableEncoder(table)
    sys.modules["densephrases.utils.open_utils"] = ou
    for name in ("densephrases.utils.kilt", "densephrases.utils.kilt.eval", "densephrases.utils.kilt.kilt_utils", "requests"):
        m = types.ModuleType(name)
        m.evaluate = m.store_data = None
        if name.endswith("kilt"):
            m.__path__ = []
        sys.modules[name] = m
    spec = importlib.util.spec_from_file_location("eval_phrase_retrieval", f"{REF}/eval_phrase_retrieval.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["eval_phrase_retrieval"] = mod
    spec.loader.exec_module(mod)
    return ref_index, mod


def main():
    docs = load_toy_docs()
    rng = np.random.default_rng(77)
    n_questions = 7
    texts = [f"what does question number {i} ask about" + ("?" if i % 2 == 0 else "") for i in range(n_questions)]
    table = {}
    ref_index, ref_eval = load_reference_eval(table)
    gold = os.path.join(ROOT, "tests", "golden")
    with tempfile.TemporaryDirectory() as tmp:
        dump_dir, idx = write_reference_layout(os.path.join(tmp, "hdf5"), docs, "toy_flat_none")
        mips = ref_index.MIPS(phrase_dump_dir=os.path.join(dump_dir, "phrase"),
                              index_path=os.path.join(dump_dir, "start", "toy_flat_none", "index.faiss"),
                              idx2id_path=os.path.join(dump_dir, "start", "toy_flat_none", "idx2id.hdf5"), cuda=False)
        q = make_queries(rng, idx.xb, n_questions)
        for t, row in zip(texts, q):
            table[t[:-1] if t.endswith("?") else t] = (row[:768].astype(np.float32), row[768:].astype(np.float32))   # load_qa_pairs strips '?'
        # gold answers: the reference's own top-1 for the even questions (exact matches), something else for the odd
        first = mips.search(q.astype(np.float64), q_texts=texts, top_k=5, aggregate=True, agg_strat="opt1")
        data = [{"id": f"toy-{i}", "question": texts[i],
                 "answers": [first[i][0]["answer"], "unrelated"] if i % 2 == 0 else ["not in the corpus"],
                 "titles": [first[i][0]["title"][0]]} for i in range(n_questions)]
        qa_path = os.path.join(gold, "eval_qa.json")
        with open(qa_path, "w") as f:
            json.dump({"data": data}, f)
        cases = []
        for top_k, agg, strat, bs in ((5, True, "opt1", 3), (3, True, "opt4", 64), (4, False, "opt1", 2)):
            args = argparse.Namespace(test_path=qa_path, do_lower_case=False, draft=False, truecase=False, cuda=False,
                                      eval_batch_size=bs, nprobe=256, top_k=top_k, max_answer_length=10, aggregate=agg,
                                      agg_strat=strat, return_sent=False, is_kilt=False, candidate_path=None, regex=False,
                                      load_dir=os.path.join(tmp, f"run{len(cases)}"), save_pred=True, eval_psg=False,
                                      verbose_logging=False, max_query_length=64)
            em1, f11, emk, f1k = ref_eval.evaluate(args, mips=mips, query_encoder=object(), tokenizer=None)
            pred_file = [p for p in os.listdir(os.path.join(args.load_dir, "pred")) if p.endswith(".pred")][0]
            with open(os.path.join(args.load_dir, "pred", pred_file)) as f:
                pred = json.load(f)
            cases.append({"top_k": top_k, "aggregate": agg, "agg_strat": strat, "eval_batch_size": bs, "pred_file": pred_file,
                          "metrics": [em1, f11, emk, f1k], "pred": pred})
    with open(os.path.join(gold, "eval_cases.json"), "w") as f:
        json.dump(cases, f)
    np.savez_compressed(os.path.join(gold, "eval_queries.npz"), texts=np.asarray(list(table.keys())),
                        vecs=np.stack([np.concatenate(table[t]) for t in table]))
    print("wrote", len(cases), "eval cases;", [c["metrics"] for c in cases])


if __name__ == "__main__":
    main()
