"""The reference's CALLERS of the hot path, loaded unmodified (source under /root/reference or byte code under
oracle/_ref), wired so that they run over whatever ``MIPS`` class the test injects:

    densephrases/options.py                 Options                                         as is
    densephrases/utils/open_utils.py        load_phrase_index, get_query2vec, load_qa_pairs  as is  (the injection point:
                                            ``from densephrases import MIPS`` at open_utils.py:8, constructed at :36-42)
    densephrases/utils/eval_utils.py        the metric functions                            as is
    densephrases/model.py                   DensePhrases (__init__, search, set_encoder, evaluate)   as is
    eval_phrase_retrieval.py                evaluate, evaluate_results, embed_all_query     as is
    train_query.py                          get_top_phrases, annotate_phrase_vecs           as is  (load_train_query below)

Stubbed, and only that: the query ENCODER and its tokenisation -- no SpanBERT weights exist offline, and the encoder is
outside the replaced path (it stays PyTorch).  ``load_encoder`` returns a placeholder, ``get_question_dataloader``
/ ``get_question_results`` (what open_utils.get_query2vec calls, open_utils.py:85-92) look the query text up in a table of
stored [1,768] start / end vectors, ``TrueCaser`` upper-cases the first letter.  Modules the callers import but this path
never executes (kilt, requests, transformers' tokenizer API) are empty stand-ins.

Test infrastructure only."""
from __future__ import annotations

import sys
import types

import numpy as np

from . import install as _install_index, load_ref_module


class _Feature:
    def __init__(self, text):
        self.tokens_ = text.split()


class _Result:
    def __init__(self, start, end):
        self.start_vec, self.end_vec = start, end


class FirstUpperCaser:
    def __init__(self, path=None):
        self.path = path

    @staticmethod
    def get_true_case(text):
        return text[:1].upper() + text[1:]


def install_callers(mips_cls, table, faiss_module=None, h5py_module=None, blosc_module=None):
    """-> (index module, open_utils module, model module, eval_phrase_retrieval module).

    ``mips_cls`` is what ``from densephrases import MIPS`` resolves to (None: the reference's own class);
    ``table``: query text -> (start fp32 [768], end fp32 [768])."""
    ref_index = _install_index(faiss_module=faiss_module, h5py_module=h5py_module, blosc_module=blosc_module)
    pkg = sys.modules["densephrases"]
    pkg.MIPS = mips_cls if mips_cls is not None else ref_index.MIPS
    pkg.Options = load_ref_module("densephrases.options").Options

    su = types.ModuleType("densephrases.utils.single_utils")
    su.load_encoder = lambda device, args, **kw: (object(), None, None)          # (model, tokenizer, config)
    su.backward_compat = lambda d: d
    sys.modules["densephrases.utils.single_utils"] = su

    sq = types.ModuleType("densephrases.utils.squad_utils")
    sq.TrueCaser = FirstUpperCaser
    # open_utils.get_query2vec (open_utils.py:85-92): dataloader, examples, features = get_question_dataloader(queries, ...)
    sq.get_question_dataloader = lambda queries, tokenizer, max_query_length, batch_size=64: (None, list(queries), [_Feature(q) for q in queries])
    sys.modules["densephrases.utils.squad_utils"] = sq

    eu = types.ModuleType("densephrases.utils.embed_utils")

    def get_question_results(question_examples, query_features, question_dataloader, device, query_encoder, batch_size=64):
        for q in question_examples:                                              # one result per query, like embed_utils.py:385-429
            s, e = table[q]
            yield _Result(np.asarray(s, np.float32)[None, :], np.asarray(e, np.float32)[None, :])
    eu.get_question_results = get_question_results
    sys.modules["densephrases.utils.embed_utils"] = eu

    for name in ("densephrases.utils.kilt", "densephrases.utils.kilt.eval", "densephrases.utils.kilt.kilt_utils", "requests"):
        m = types.ModuleType(name)
        m.evaluate = m.store_data = None
        if name.endswith("kilt"):
            m.__path__ = []
        sys.modules[name] = m
    if "transformers" not in sys.modules:
        # open_utils.py:13-18 imports four names it only uses in load_cross_encoder (not on this path); the installed
        # transformers (5.x) takes seconds to import and has dropped some of them
        tr = types.ModuleType("transformers")
        tr.MODEL_MAPPING = tr.AutoConfig = tr.AutoTokenizer = tr.AutoModel = None
        sys.modules["transformers"] = tr
        stub_tr = True
    else:
        stub_tr = False
    try:
        ou = load_ref_module("densephrases.utils.open_utils")
    finally:
        if stub_tr:
            sys.modules.pop("transformers", None)
    model = load_ref_module("densephrases.model")
    pkg.DensePhrases = model.DensePhrases
    ev = load_ref_module("eval_phrase_retrieval")
    return ref_index, ou, model, ev


def load_train_query():
    """The reference's train_query.py (the SECOND caller of MIPS.search: query-side fine-tuning, train_query.py:182-275), loaded
    unmodified after install_callers.  Its imports this path never executes -- transformers' AdamW / scheduler (:24-27; AdamW is gone
    from the installed transformers 5.x), requests -- are empty names while the file loads."""
    stub = types.ModuleType("transformers")
    stub.AdamW = stub.get_linear_schedule_with_warmup = None
    saved = sys.modules.get("transformers")
    sys.modules["transformers"] = stub
    try:
        return load_ref_module("train_query")
    finally:
        if saved is not None:
            sys.modules["transformers"] = saved
        else:
            sys.modules.pop("transformers", None)


def uninstall():
    """drop every module install_callers / refshim.install registered (tests restore sys.modules)"""
    for name in list(sys.modules):
        if name == "densephrases" or name.startswith("densephrases.") or name in ("eval_phrase_retrieval", "train_query", "faiss", "h5py", "blosc",
                                                                                 "spacy", "spacy.lang", "spacy.lang.en", "ujson", "requests"):
            sys.modules.pop(name, None)
