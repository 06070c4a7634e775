"""The user-facing façade (densephrases_amd.model.DensePhrases) against the outputs of the reference's own
model.py ``DensePhrases.search`` run over the reference's own MIPS (oracle/make_golden_model.py): all four retrieval
units, top_k doubling, truecasing, single-string queries, the unsupported-unit error."""
import json
import os

import numpy as np
import pytest

from tests._golden import GOLD, load_toy_docs

pytestmark = pytest.mark.gpu

CASES = json.load(open(os.path.join(GOLD, "model_cases.json")))
_Z = np.load(os.path.join(GOLD, "model_queries.npz"))
TABLE = {str(t): v for t, v in zip(_Z["texts"].tolist(), _Z["vecs"])}


class _TableEncoder:
    def __call__(self, queries):
        out = []
        for q in queries:
            v = TABLE[q] if q in TABLE else TABLE[q[:1].lower() + q[1:]]
            out.append((v[None, :768].tolist(), v[None, 768:].tolist(), q.split()))
        return out


class _FirstUpper:
    @staticmethod
    def get_true_case(text):
        return text[:1].upper() + text[1:]


@pytest.fixture(scope="module")
def facade():
    from densephrases_amd import DocMeta, DocStore, MIPS
    from densephrases_amd.model import DensePhrases
    store = DocStore([DocMeta(m.doc_idx, m.title, m.context, m.f2o_start, m.word2char_start, m.word2char_end, m.start)
                      for m in load_toy_docs()])
    return DensePhrases.from_parts(MIPS.from_store(store), _TableEncoder(), _FirstUpper())


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_facade_matches_the_reference_model_py(facade, ci):
    c = CASES[ci]
    if "error" in c:
        with pytest.raises(NotImplementedError) as e:
            facade.search(query=list(TABLE.keys()), retrieval_unit=c["retrieval_unit"])
        assert str(e.value) == c["message"]
        return
    retrieved, rets = facade.search(query=c["query"], retrieval_unit=c["retrieval_unit"], top_k=c["top_k"],
                                    truecase=c["truecase"], return_meta=True)
    assert retrieved == c["retrieved"]
    got = [rets] if c["single"] else rets
    want = [c["meta"]] if c["single"] else c["meta"]
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert len(g) == len(w)
        for a, b in zip(g, w):
            for key in ("context", "title", "doc_idx", "start_pos", "end_pos", "start_idx", "end_idx", "answer"):
                assert a[key] == b[key], (key, a[key], b[key])
            assert np.isclose(a["score"], b["score"], rtol=1e-6, atol=1e-4)


def test_facade_takes_the_query_tensor_on_the_device(facade):
    """a query2vec that returns a [B, 1536] GPU tensor (the encoder's output as it is) goes through search_device"""
    import torch
    from densephrases_amd.model import DensePhrases
    c = next(c for c in CASES if c.get("retrieval_unit") == "phrase" and not c["single"] and not c["truecase"])
    dev_enc = lambda qs: torch.tensor(np.stack([TABLE[q] for q in qs])).cuda()        # noqa: E731
    f2 = DensePhrases.from_parts(facade.mips, dev_enc, None)
    assert f2.search(query=c["query"], retrieval_unit="phrase", top_k=c["top_k"], truecase=False) == c["retrieved"]

