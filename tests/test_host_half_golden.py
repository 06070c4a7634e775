"""The python (host) half of the product's MIPS.search_phrase -- interleave, metadata lookup, dict assembly, answer
slice, paragraph / sentence cropping, per-query sort, aggregation -- against outputs of the reference's own index.py
(tests/golden), on CPU: the device-stage values it consumes (top-k ids / scores, window arg-max results) come from the
oracle here, on a GPU box from libdph (tests/test_gpu_search.py runs the same goldens end to end)."""
import numpy as np
import pytest

from oracle import mips_oracle as O
from tests._golden import compare_results, load_cases, load_toy_docs, load_toy_index

CASES, VECS = load_cases()


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_assemble_and_aggregate_match_reference_golden(ci):
    c = CASES[ci]
    if c["return_idxs"] and c["branch"] == "hdf5":
        pytest.skip("reference HDF5 branch returns raw int8 for the candidate's own vector (index.py:263-272); "
                    "the product follows the RAM branch")
    from densephrases_amd import DocMeta, DocStore
    from densephrases_amd.index import MIPS
    oidx = load_toy_index()
    docs = load_toy_docs()
    m = MIPS.__new__(MIPS)                      # the host half needs no shard
    m.store = DocStore([DocMeta(d.doc_idx, d.title, d.context, d.f2o_start, d.word2char_start, d.word2char_end, d.start)
                        for d in docs])
    m.num_docs_list = []
    q = c["query_arr"].astype(np.float32)
    B, k, L = q.shape[0], c["top_k"], c["L"]
    sdoc, sword, sI, edoc, eword, eI, sD, eD = O.search_dense(oidx, q, k)
    flat = lambda a: np.reshape(np.asarray(a), [-1])          # noqa: E731
    qq = np.repeat(q, k, axis=0)
    qs, qe = np.split(qq, 2, axis=1)
    pred_end, best1, _, end_vecs, am1 = O.window_rescore(oidx, qe, flat(sdoc), flat(sword), flat(sI), flat(sD), L, "end", "ram")
    pred_start, best2, _, start_vecs, am2 = O.window_rescore(oidx, qs, flat(edoc), flat(eword), flat(eI), flat(eD), L, "start", "ram")
    v1 = v2 = None
    if c["return_idxs"]:
        n = B * k
        # libdph's layout: [:,0] the candidate's own row, [:,1] the arg-max slot's row (dph_window.hip)
        v1 = (end_vecs[:, 0, :], end_vecs[np.arange(n), am1])
        v2 = (start_vecs[:, -1, :], start_vecs[np.arange(n), am2])
    outs = m._assemble(B, k, flat(sdoc), flat(sword), flat(edoc), flat(eword), np.asarray(pred_end), np.asarray(best1),
                       np.asarray(pred_start), np.asarray(best2), v1, v2, c["return_sent"])
    if c["aggregate"]:
        outs = [m.aggregate_results(r, k, f"q{i}", c["agg_strat"]) for i, r in enumerate(outs)]
    compare_results(outs, c["results"], VECS, case=c)


@pytest.mark.parametrize("seed", range(12))
def test_cpp_assemble_equals_the_python_restatement(seed):
    """csrc/dph_host.cpp (what MIPS runs) against ``MIPS._assemble_py`` (index.py:373-421 line by line) on generated
    batches: non-ASCII contexts (positions are code points), padding candidates (doc -1), masked windows (score -1e9),
    missing end predictions (-1), equal scores (stable order), with and without sentence cropping and vectors."""
    from densephrases_amd import DocMeta, DocStore
    from densephrases_amd.index import MIPS
    rng = np.random.default_rng(seed)
    # incl. the widened sentence rule: closing quotes / brackets after the terminator, full-width and CJK terminators, a
    # terminator inside a token ("3.5", "e.g.x"), doubled terminators
    words = ["alpha", "běta", "γάμμα", "delta.", "Эпсилон!", "zeta?", "η", "theta", "iota.", "κάππα", 'said."', "(end.)", "了。",
             "３！", "3.5", "e.g.x", "what?!", "wow!”", "a.b"]
    docs = []
    for d in range(5):
        pars, pos, w2cs, w2ce = [], 0, [], []
        n_par = int(rng.integers(1, 4))
        for pi in range(n_par):
            toks = [words[int(i)] for i in rng.integers(0, len(words), int(rng.integers(3, 12)))]
            for wi, w in enumerate(toks):
                w2cs.append(pos)
                w2ce.append(pos + len(w))
                pos += len(w) + (1 if wi < len(toks) - 1 else 0)
            pars.append(" ".join(toks))
            pos += len(" [PAR] ")
        ctx = " [PAR] ".join(pars)
        n_tok = len(w2cs)
        keep = np.sort(rng.choice(n_tok, max(1, int(n_tok * 0.7)), replace=False)).astype(np.int64)
        docs.append(DocMeta(100 + d, f"Títle {d}", ctx, keep, np.asarray(w2cs, np.int32), np.asarray(w2ce, np.int32),
                            np.zeros((len(keep), 768), np.int8)))
    m = MIPS.__new__(MIPS)
    m.store = DocStore(docs)
    B, k = 3, 4
    n = B * k

    def cand():
        doc = rng.integers(0, 5, n)
        d_ids = np.array([docs[i].doc_idx for i in doc])
        w = np.array([int(rng.integers(0, len(docs[i].f2o_start))) for i in doc])
        w2 = np.array([int(rng.integers(wi, len(docs[i].f2o_start))) for i, wi in zip(doc, w)])
        return d_ids, w, w2

    sdoc, sword, pend = cand()
    edoc, pstart, eword = cand()
    best1 = np.round(rng.normal(0, 3, n), 1)               # rounded: equal scores occur -> the stable order matters
    best2 = np.round(rng.normal(0, 3, n), 1)
    sdoc[1] = -1                                           # FAISS padding -> dummy
    best2[2] = -1e9 + 3.0                                  # every window masked out
    pend[3] = -1                                           # no valid end found
    best1[3] = -1e9 + 1.0
    for ridx in (False, True):
        v1 = v2 = None
        if ridx:
            v1 = (rng.normal(size=(n, 768)).astype(np.float32), rng.normal(size=(n, 768)).astype(np.float32))
            v2 = (rng.normal(size=(n, 768)).astype(np.float32), rng.normal(size=(n, 768)).astype(np.float32))
        for sent in (False, True):
            args = (B, k, sdoc, sword, edoc, eword, pend.astype(np.int32), best1, pstart.astype(np.int32), best2, v1, v2, sent)
            got, want = m._assemble(*args), m._assemble_py(*args)
            assert len(got) == len(want) == B
            for g, w in zip(got, want):
                assert len(g) == len(w)
                for a, b in zip(g, w):
                    assert list(a.keys()) == list(b.keys())
                    for key in b:
                        if key in ("start_vec", "end_vec"):
                            assert (a[key] is None and b[key] is None) or np.array_equal(a[key], b[key])
                        else:
                            assert a[key] == b[key] and type(a[key]) is type(b[key]), (key, a[key], b[key])


@pytest.mark.parametrize("strat", ["opt1", "opt2", "opt3", "opt4"])
def test_cpp_aggregate_equals_the_python_restatement(strat):
    """csrc/dph_host.cpp: aggregate against ``MIPS._aggregate_results_py`` (index.py:424-448) on inputs with many key
    collisions, equal scores (stable order), already merged multi-title lists and, for opt4, title merging."""
    import copy
    from densephrases_amd.index import MIPS
    rng = np.random.default_rng(hash(strat) % 1000)
    m = MIPS.__new__(MIPS)
    for trial in range(30):
        res = []
        for _ in range(int(rng.integers(0, 25))):
            t = f"T{int(rng.integers(0, 4))}"
            res.append({"title": [t] if rng.random() < 0.8 else [t, "X_1_2"], "context": f"ctx {int(rng.integers(0, 5))}",
                        "start_pos": int(rng.integers(0, 3)), "end_pos": int(rng.integers(3, 6)),
                        "answer": ["The Cat", "cat", "a cat!", "dog", "Dog."][int(rng.integers(0, 5))],
                        "score": float(np.round(rng.normal(0, 2), 1))})
        a, b = copy.deepcopy(res), copy.deepcopy(res)
        got = m.aggregate_results(a, 10, "q", strat)
        want = MIPS._aggregate_results_py(b, 10, "q", strat)
        assert got == want
        assert a == b                                   # the in-place effects (scores, merged titles) are the same too
    with pytest.raises(NotImplementedError):
        m.aggregate_results([], 10, "q", "opt9")


def test_return_sent_goldens_reassemble_from_their_own_candidates():
    """Every result dict of the ``return_sent`` goldens (toy dump and the OPQ-IVFPQ index: the second holds answers that cross
    a paragraph mark, where the restated spaCy rule leaves the opening bracket of ``[PAR]`` with the sentence that ended)
    is rebuilt from its own (doc, start word, end word, score) by the C++ host half and by the python restatement:
    context, positions and answer must come out as the reference's ``adjust`` + ``adjust_sent`` left them."""
    import json
    import os
    from densephrases_amd import DocMeta, DocStore
    from densephrases_amd.index import MIPS
    from tests._golden import GOLD
    docs = load_toy_docs()
    m = MIPS.__new__(MIPS)
    m.store = DocStore([DocMeta(d.doc_idx, d.title, d.context, d.f2o_start, d.word2char_start, d.word2char_end, d.start)
                        for d in docs])
    m.num_docs_list = []
    pq_cases = json.load(open(os.path.join(GOLD, "pq_cases.json")))
    # ... and the facade's retrieval_unit = "sentence" goldens (model.py:82-87 searches with return_sent=True)
    model_cases = [{"return_sent": True, "results": [c["meta"]] if c["single"] else c["meta"]}
                   for c in json.load(open(os.path.join(GOLD, "model_cases.json"))) if c.get("retrieval_unit") == "sentence"]
    assert model_cases
    checked = crossing = 0
    for c in list(CASES) + pq_cases + model_cases:
        if not c["return_sent"]:
            continue
        for res in c["results"]:
            k = len(res)
            if k == 0:
                continue
            sdoc = np.array([r["doc_idx"] for r in res])
            sword = np.array([r["start_idx"] for r in res])
            pend = np.array([r["end_idx"] for r in res], np.int32)
            best1 = np.array([r["score"] for r in res], np.float64)
            args = (1, k, sdoc, sword, np.full(k, -1), np.zeros(k, np.int64), pend, best1, np.full(k, -1, np.int32),
                    np.full(k, -1e9), None, None, True)
            for out in (m._assemble(*args)[0], m._assemble_py(*args)[0]):
                assert len(out) == k
                for a, g in zip(out, res):
                    for key in ("context", "start_pos", "end_pos", "answer", "doc_idx", "start_idx", "end_idx"):
                        assert a[key] == g[key], (key, a[key], g[key])
            checked += k
            crossing += sum("[ PAR]" in r["context"] for r in res)
    assert checked >= 60 and crossing >= 2


def test_doc_cache_generations_keep_a_working_set_and_fused_aggregate_equals_two_calls():
    """The C++ host half keeps documents in two generations: a working set that fits the capacity is fetched once however
    the batches alternate (round 3 cleared the whole cache when it filled: two alternating batches re-fetched every document
    on every batch), a larger one is fetched again but the results never change; ``assemble(..., agg_strat)`` = ``assemble``
    followed by ``aggregate`` on every query's list."""
    import copy
    from densephrases_amd import _dph_host
    from densephrases_amd.index import normalize_answer
    from densephrases_amd.synth import SynthDocStore
    store = SynthDocStore()
    B, k = 4, 5
    n = 2 * B * k

    def batch(seed, n_docs):
        r = np.random.default_rng(seed)
        doc = r.integers(0, n_docs, n).astype(np.int64) + 1000 * seed
        s = r.integers(0, 90, n).astype(np.int64)
        e = s + r.integers(0, 5, n)
        sc = np.round(r.normal(0, 3, n), 1)
        return doc, s, e.astype(np.int64), sc

    big = _dph_host.HostHalf(lambda d: store.doc_meta(int(d)), 1 << 16)
    want = {}
    for seed in range(6):
        d, s, e, sc = batch(seed, 30)
        plain = big.assemble(B, k, d, s, e, sc, None, None, False)
        want[seed] = [_dph_host.aggregate(copy.deepcopy(r), "opt3", normalize_answer) for r in plain]
    # capacity 128 = generations of 64: two alternating batches (<= 60 documents together) are fetched exactly once
    h = _dph_host.HostHalf(lambda d: store.doc_meta(int(d)), 128)
    for it in range(10):
        seed = it % 2
        d, s, e, sc = batch(seed, 30)
        assert h.assemble(B, k, d, s, e, sc, None, None, False, "opt3", normalize_answer) == want[seed]
    distinct = len(set(batch(0, 30)[0].tolist()) | set(batch(1, 30)[0].tolist()))
    assert h.fetched_docs() == distinct and h.cached_docs() == distinct
    # a working set beyond the capacity (6 batches of ~25 documents against generations of 16): re-fetched, never wrong, bounded
    h = _dph_host.HostHalf(lambda d: store.doc_meta(int(d)), 32)
    for it in range(18):
        seed = it % 6
        d, s, e, sc = batch(seed, 30)
        assert h.assemble(B, k, d, s, e, sc, None, None, False, "opt3", normalize_answer) == want[seed]
        assert h.cached_docs() <= 2 * 40
    assert h.fetched_docs() > 6 * 20


@pytest.mark.parametrize("k,sent", [(5, False), (5, True), (70, False), (70, True)])
def test_fused_native_aggregate_equals_assemble_then_aggregate(k, sent):
    """``assemble(..., agg_strat)`` decides MIPS.aggregate_results' survivors in C++ BEFORE any python object exists (opt1 / opt2 /
    opt3: candidates compared field by field, or through a key string when 2 * top_k > 128 or the context is joined sentences) and
    only those become dicts; the result must be what ``assemble`` followed by the general ``aggregate`` gives -- on documents that
    share a title and documents that share their whole text (different doc_idx, same key strings), equal scores, dummies, and a
    batch large enough to go through the worker threads."""
    import copy
    from densephrases_amd import DocMeta, _dph_host
    from densephrases_amd.index import normalize_answer
    rng = np.random.default_rng(k + sent)
    words = ["alpha", "Beta.", "gamma", "delta!", "The", "epsilon?", "zeta", "eta."]
    docs = {}
    for d in range(40):
        src = d if d % 4 else max(d - 4, 0)                  # every fourth document repeats an earlier one's text (and title)
        r = np.random.default_rng(1000 + src)
        pars, pos, w2cs, w2ce = [], 0, [], []
        for pi in range(int(r.integers(1, 4))):
            toks = [words[int(i)] for i in r.integers(0, len(words), int(r.integers(4, 10)))]
            for wi, w in enumerate(toks):
                w2cs.append(pos)
                w2ce.append(pos + len(w))
                pos += len(w) + (1 if wi < len(toks) - 1 else 0)
            pars.append(" ".join(toks))
            pos += len(" [PAR] ")
        docs[d] = DocMeta(d, f"Title {src % 7}", " [PAR] ".join(pars), np.arange(len(w2cs), dtype=np.int64), np.asarray(w2cs, np.int32),
                          np.asarray(w2ce, np.int32))
    B = 130 if k == 70 else 7                                # 2 * 130 * 70 = 18 200 candidates: the threaded path
    n = 2 * B * k
    doc = rng.integers(0, 40, n).astype(np.int64)
    s = np.array([int(rng.integers(0, len(docs[int(d)].f2o_start))) for d in doc], np.int64)
    e = np.array([int(rng.integers(si, min(si + 4, len(docs[int(d)].f2o_start)))) for d, si in zip(doc, s)], np.int64)
    sc = np.round(rng.normal(0, 3, n), 1)
    twin = rng.integers(0, n // 2, n // 6) * 2            # the same span found twice by a query (start- and end-candidate)
    doc[twin + 1], s[twin + 1], e[twin + 1] = doc[twin], s[twin], e[twin]
    doc[rng.integers(0, n, 9)] = -1
    sc[rng.integers(0, n, 9)] = -1e9
    h = _dph_host.HostHalf(lambda d: docs[int(d)], 1 << 12)
    plain = h.assemble(B, k, doc, s, e, sc, None, None, sent)
    for strat in ("opt1", "opt2", "opt3", "opt4"):
        want = [_dph_host.aggregate(copy.deepcopy(r), strat, normalize_answer) for r in plain]
        got = h.assemble(B, k, doc, s, e, sc, None, None, sent, strat, normalize_answer)
        assert got == want, strat
        assert sum(len(r) for r in got) < sum(len(r) for r in plain)            # something WAS de-duplicated
    with pytest.raises(TypeError):
        h.assemble(B, k, doc, s, e, sc, None, None, sent, "opt9", normalize_answer)


@pytest.mark.parametrize("agg,sent", [(None, False), ("opt1", False), ("opt3", True), ("opt4", False)])
def test_prepare_async_from_a_record_equals_assemble(agg, sent):
    """HostHalf.prepare_async: the record the GPU half leaves in pinned memory (I, best, pred, status) -> ids to (doc, word) through
    the dph_id2docword entry point it is given, start / end candidates interleaved as MIPS.search_phrase does, document cache, phase 1,
    all on a worker thread; ``materialize`` then builds the dicts.  Must equal ``assemble`` over the arrays python would have built;
    a batch with an uncertified row reports ``needs_exact`` and refuses to materialize; two batches may be in flight."""
    import ctypes
    from densephrases_amd import _dph_host
    from densephrases_amd.index import normalize_answer
    from densephrases_amd.synth import SynthDocStore
    store = SynthDocStore()
    B, k = 9, 6
    rng = np.random.default_rng(3)

    @ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_int64, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32))
    def id2docword(handle, I, n, doc, word):                    # documents of 100 tokens, like the synthetic dumps; -1 stays unknown
        for i in range(n):
            doc[i], word[i] = (I[i] // 100, I[i] % 100) if I[i] >= 0 else (-1, -1)
        return 0

    fn = ctypes.cast(id2docword, ctypes.c_void_p).value

    def record(seed):
        r = np.random.default_rng(seed)
        I = r.integers(0, 4000, (2 * B, k)).astype(np.int64)
        I[1, 2] = -1                                               # FAISS padding
        word = I % 100
        pred = np.where(r.random((2 * B, k)) < 0.9, np.clip(word + r.integers(-3, 4, word.shape), 0, 99), -1).astype(np.int32)
        pred[:B] = np.maximum(pred[:B], word[:B].astype(np.int32) * (pred[:B] >= 0) - (pred[:B] < 0))      # end >= start for start candidates
        pred[B:] = np.where(pred[B:] >= 0, np.minimum(pred[B:], word[B:]), -1)                             # start <= end for end candidates
        best = np.round(r.normal(0, 3, (2 * B, k)), 1)
        best[3, 1] = -1e9
        best[B:][pred[B:] < 0] = -1e9                               # no valid start found: the window kernel masks the candidate out
        best[I < 0] = -1e9
        return I, best, pred, np.zeros(2 * B, np.int32)

    def python_arrays(I, best, pred):
        doc, word = I // 100, I % 100
        doc = np.where(I >= 0, doc, -1)
        word = np.where(I >= 0, word, -1)
        flat = lambda a: a.reshape(-1)                              # noqa: E731
        doc_i = np.stack([flat(doc[:B]), flat(doc[B:])], 1).reshape(-1)
        start_i = np.stack([flat(word[:B]), flat(pred[B:]).astype(np.int64)], 1).reshape(-1)
        end_i = np.stack([flat(pred[:B]).astype(np.int64), flat(word[B:])], 1).reshape(-1)
        score_i = np.stack([flat(best[:B]), flat(best[B:])], 1).reshape(-1)
        return doc_i, start_i, end_i, score_i

    h = _dph_host.HostHalf(store.doc_meta, 1 << 12)
    recs = [record(s) for s in (1, 2)]
    preps = [h.prepare_async(B, k, I.ctypes.data, best.ctypes.data, pred.ctypes.data, st.ctypes.data, fn, 1, sent, agg) for I, best, pred, st in recs]
    for (I, best, pred, st), P in zip(recs, preps):
        needs_exact, num_docs = P.wait()
        assert needs_exact is False
        d = np.where(I >= 0, I // 100, -1)
        want_docs = np.mean([len(set(d[q].tolist()) | set(d[B + q].tolist())) for q in range(B)])
        assert abs(num_docs - want_docs) < 1e-9
        got = h.materialize(P, normalize_answer)
        want = _dph_host.HostHalf(store.doc_meta, 1 << 12).assemble(B, k, *python_arrays(I, best, pred), None, None, sent, agg, normalize_answer)
        assert got == want and sum(len(r) for r in got) > B
    I, best, pred, st = record(5)
    st[4] = 1                                                       # an uncertified row
    P = h.prepare_async(B, k, I.ctypes.data, best.ctypes.data, pred.ctypes.data, st.ctypes.data, fn, 1, sent, agg)
    assert P.wait()[0] is True
    with pytest.raises(RuntimeError):
        h.materialize(P, normalize_answer)
    st[4] = 3                                                       # a non-finite row is final: no repair asked for
    P = h.prepare_async(B, k, I.ctypes.data, best.ctypes.data, pred.ctypes.data, st.ctypes.data, fn, 1, sent, agg)
    assert P.wait()[0] is False and len(h.materialize(P, normalize_answer)) == B
