#!/usr/bin/env python3
"""Generate tests/golden/model_cases.json by running the REFERENCE's own ``DensePhrases.search`` façade
(/root/reference/densephrases/model.py:55-109, loaded unmodified) over the reference's own ``MIPS``
(/root/reference/densephrases/index.py through oracle/refshim) on the toy dump of tests/golden/toy_dump.npz.

What is stubbed -- and only that: the query encoder.  ``model.py`` imports ``Options``, ``load_encoder``,
``load_phrase_index``, ``get_query2vec``, ``load_qa_pairs`` and ``TrueCaser`` at module level (model.py:6-9); those
names are provided by empty stand-ins because ``DensePhrases.__init__`` (which would download SpanBERT) is bypassed:
the object is created with ``object.__new__`` and given ``.mips`` (the reference MIPS), ``.query2vec`` (a table
look-up: query text -> the stored [1,768] start / end vectors of tests/golden/model_queries.npz, returned in the
``(start.tolist(), end.tolist(), tokens)`` shape of open_utils.py:83-101) and ``.truecase`` (identity for mixed-case
text, upper-cases the first letter of all-lower-case queries so the ``truecase`` branch of model.py:67-68 is taken).

Run from the repo root in the build container (needs /root/reference):   python -m oracle.make_golden_model
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import refshim                                  # noqa: E402
from oracle.make_golden import write_reference_layout       # noqa: E402
from oracle.synth_dump import make_queries                  # noqa: E402
from tests._golden import load_toy_docs                     # noqa: E402


def load_reference_model():
    ref_index = refshim.install()
    pkg = sys.modules["densephrases"]
    pkg.Options = type("Options", (), {})
    pkg.MIPS = ref_index.MIPS
    utils = types.ModuleType("densephrases.utils")
    utils.__path__ = [f"{refshim.REFERENCE_ROOT}/densephrases/utils"]
    sys.modules["densephrases.utils"] = utils
    for name, attrs in (("single_utils", ["load_encoder"]), ("open_utils", ["load_phrase_index", "get_query2vec", "load_qa_pairs"]),
                        ("squad_utils", ["TrueCaser"])):
        m = types.ModuleType(f"densephrases.utils.{name}")
        for a in attrs:
            setattr(m, a, None)
        sys.modules[f"densephrases.utils.{name}"] = m
    spec = importlib.util.spec_from_file_location("densephrases.model", f"{refshim.REFERENCE_ROOT}/densephrases/model.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["densephrases.model"] = mod
    spec.loader.exec_module(mod)
    return ref_index, mod


class TableEncoder:
    """query text -> stored vectors, in the output shape of open_utils.get_query2vec (open_utils.py:83-101)"""

    def __init__(self, table):
        self.table = table

    def __call__(self, queries):
        return [(self.table[q][0][None, :].tolist(), self.table[q][1][None, :].tolist(), q.split()) for q in queries]


class FirstUpperCaser:
    @staticmethod
    def get_true_case(text):
        return text[:1].upper() + text[1:]


def main():
    ref_index, ref_model = load_reference_model()
    docs = load_toy_docs()
    rng = np.random.default_rng(2026)
    texts = ["who is alpha", "where was Bravo born", "what is the capital of charlie", "when did Delta Echo happen", "golf hotel"]
    cases, qstore = [], {}
    with tempfile.TemporaryDirectory() as tmp:
        dump_dir, idx = write_reference_layout(os.path.join(tmp, "hdf5"), docs, "toy_flat_none")
        mips = ref_index.MIPS(phrase_dump_dir=os.path.join(dump_dir, "phrase"),
                              index_path=os.path.join(dump_dir, "start", "toy_flat_none", "index.faiss"),
                              idx2id_path=os.path.join(dump_dir, "start", "toy_flat_none", "idx2id.hdf5"), cuda=False)
        q = make_queries(rng, idx.xb, len(texts))
        table = {}
        for t, row in zip(texts, q):
            for key in (t, FirstUpperCaser.get_true_case(t)):          # the truecased text reaches query2vec (model.py:67-71)
                table[key] = (row[:768].astype(np.float32), row[768:].astype(np.float32))
            qstore[t] = row.astype(np.float32)
        dp = object.__new__(ref_model.DensePhrases)
        dp.mips, dp.query2vec, dp.truecase = mips, TableEncoder(table), FirstUpperCaser()
        for unit in ("phrase", "sentence", "paragraph", "document"):
            for top_k, truecase, single in ((5, True, False), (3, False, False), (4, True, True)):
                query = texts[1] if single else list(texts)
                retrieved, rets = ref_model.DensePhrases.search(dp, query=query, retrieval_unit=unit, top_k=top_k,
                                                                truecase=truecase, return_meta=True)
                keep = ("context", "title", "doc_idx", "start_pos", "end_pos", "start_idx", "end_idx", "score", "answer")

                def slim(rs):
                    return [{k: (float(r[k]) if k == "score" else (int(r[k]) if isinstance(r[k], (np.integer,)) else r[k]))
                             for k in keep} for r in rs]
                cases.append({"retrieval_unit": unit, "top_k": top_k, "truecase": truecase, "single": single, "query": query,
                              "retrieved": retrieved, "meta": slim(rets) if single else [slim(r) for r in rets]})
        try:
            ref_model.DensePhrases.search(dp, query=list(texts), retrieval_unit="passage")
            raise AssertionError("the reference accepts only phrase / sentence / paragraph / document")
        except NotImplementedError as e:
            cases.append({"retrieval_unit": "passage", "error": "NotImplementedError", "message": str(e)})
    gold = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(gold, "model_cases.json"), "w") as f:
        json.dump(cases, f)
    np.savez_compressed(os.path.join(gold, "model_queries.npz"), texts=np.asarray(texts),
                        vecs=np.stack([qstore[t] for t in texts]))
    print(f"wrote {len(cases)} façade cases")


if __name__ == "__main__":
    main()
