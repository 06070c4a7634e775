"""Non-finite and extreme query rows.  Contract (include/dph.h, dph_search): a query row with a NaN / Inf element
is answered like FAISS' flat search answers it -- ids -1, scores -FLT_MAX -- with status DPH_ROW_NONFINITE (3) in the device forms
and a count in dph_search_stats.nonfinite; every other row of the call is answered as ever, the call returns DPH_OK, and the row
costs what an ordinary row costs (its stand-in is searched and certified; before round 6 a NaN poisoned the certificate's bounds and
the row walked the whole retry chain into fp64 full scans until the batch raised).  Finite rows of any magnitude -- 1e30, all
zeros -- are searched exactly.  Reference: index.py:195-200 hands FAISS whatever the encoder produced (fp16 encoders overflow)."""
import numpy as np
import pytest

from oracle import mips_oracle as O

pytestmark = pytest.mark.gpu

FLT_MAX = np.float32(3.4028234663852886e38)


def _queries(rng, xb, n=9):
    x = rng.normal(0, 0.5, (n, 768)).astype(np.float32)
    x[0] = xb[17].astype(np.float32) / 20 - 2
    x[1, 5] = np.nan
    x[2, 700] = np.inf
    x[3, :] = -np.inf
    x[4] *= np.float32(1e30)                       # finite, huge: exact
    x[5] = 0.0                                     # all scores equal: the k lowest ids
    x[6, 0], x[6, 767] = np.nan, np.inf
    bad = np.array([1, 2, 3, 6])
    good = np.array([0, 4, 5, 7, 8])
    return x, bad, good


def test_flat_search_host_and_device_forms():
    import torch
    from densephrases_amd import Shard
    rng = np.random.default_rng(5)
    xb = O.float_to_int8(rng.standard_normal((50000, 768), dtype=np.float32) * np.float32(0.6))
    s = Shard(xb.shape[0], device=0, id_base=100)
    s.upload(xb)
    s.finalize()
    x, bad, good = _queries(rng, xb)
    k = 10
    D, I = s.search(x, k)
    st = s.stats()
    assert st["nonfinite"] == len(bad) and st["uncertified"] == 0 and st["rows"] == len(x)
    assert (I[bad] == -1).all() and (D[bad] == -FLT_MAX).all()
    Dr, Ir, D64 = O.flat_ip_search(x[good], xb, k, id_base=100)
    ok, msg = O.topk_equivalent(D[good], I[good], D64, Ir)
    assert ok, msg
    assert I[0, 0] == 117 and list(I[5]) == list(range(100, 100 + k)) and (D[5] == 0).all()
    assert st["exact_fallback"] <= 1                          # (the zero row may need it; a NaN row never does)
    # device form: status 3 for the flagged rows, 0 for the others; a batch of ONLY non-finite rows
    dev = torch.device("cuda", 0)
    for xs, want_bad in ((x, bad), (x[bad], np.arange(len(bad)))):
        xd = torch.from_numpy(xs).to(dev)
        Dd = torch.empty((len(xs), k), dtype=torch.float32, device=dev)
        Id = torch.empty((len(xs), k), dtype=torch.int64, device=dev)
        sd = torch.empty(len(xs), dtype=torch.int32, device=dev)
        s.search_dev(xd.data_ptr(), len(xs), k, Dd.data_ptr(), Id.data_ptr(), sd.data_ptr())
        torch.cuda.synchronize()
        status = sd.cpu().numpy()
        assert (status[want_bad] == 3).all() and (np.delete(status, want_bad) == 0).all()
        assert (Id.cpu().numpy()[want_bad] == -1).all() and (Dd.cpu().numpy()[want_bad] == -FLT_MAX).all()
        if len(xs) == len(x):
            np.testing.assert_array_equal(Id.cpu().numpy()[good], I[good])
        assert s.stats()["nonfinite"] == len(want_bad)
    s.close()


def test_pq_search_flags_the_row_and_does_not_fail_over():
    from densephrases_amd import faiss_io as F
    from oracle import ivfpq_oracle as P
    from tests.test_pq import _index_from_list_numbers, _same_topk, _shard
    rng = np.random.default_rng(9)
    nlist, M = 65536, 96
    cent = rng.normal(0, 0.5, (nlist, 768)).astype(np.float32)
    ix, A = _index_from_list_numbers(rng, nlist, M, rng.integers(0, nlist, 9000), cent)
    s = _shard(ix)
    x = rng.normal(0, 0.5, (7, 768)).astype(np.float32)
    x[2, 9] = np.nan
    x[4, :] = np.inf
    good = np.array([0, 1, 3, 5, 6])
    D, I = s.search_ivf(x, 10, 256)
    assert (I[[2, 4]] == -1).all() and (D[[2, 4]] == -FLT_MAX).all()
    Dr, Ir = P.search(ix, x[good], 10, 256)
    _same_topk(D[good], I[good], Dr, Ir)
    failed_over, _ = s.debug_pq_coarse()
    assert failed_over is False                               # (a NaN estimate used to send the whole pass down the bf16x3 chain)
    st = s.stats()
    assert st["nonfinite"] == 2 and st["uncertified"] == 0
    s.close()


def test_mips_search_answers_the_other_queries_and_leaves_the_flagged_one_empty(monkeypatch):
    from densephrases_amd import DocMeta, DocStore, MIPS
    from oracle.synth_dump import make_dump, make_queries
    docs = make_dump(seed=3, n_docs=40, d=768)
    store = DocStore([DocMeta(m.doc_idx, m.title, m.context, m.f2o_start, m.word2char_start, m.word2char_end, m.start) for m in docs])
    mips = MIPS.from_store(store)
    rng = np.random.default_rng(1)
    q = make_queries(rng, store.rows, 6)
    kw = dict(top_k=5, aggregate=True, agg_strat="opt1")
    want = mips.search(q, q_texts=list("abcdef"), **kw)
    qn = q.copy()
    qn[2, 100] = np.nan                                        # start half
    qn[4, 768 + 3] = np.inf                                    # end half only: the start half's candidates still come back
    got = mips.search(qn, q_texts=list("abcdef"), **kw)
    for i in (0, 1, 3, 5):
        assert got[i] == want[i]
    assert got[2] == [] or all(r["score"] > -1e5 for r in got[2])
    # every result of the half-flagged query comes from the finite half's candidates
    assert len(got[4]) <= len(want[4]) + 5
    streamed = list(mips.search_stream([qn, q], q_texts=[list("abcdef")] * 2, **kw))
    assert streamed[0][0] == want[0] and streamed[1] == want
    # the same three batches deep (what a stream of large batches runs: the interpreter-free part of a batch's host half on a worker
    # thread, from the record in pinned memory, while this thread builds the previous batch's dicts) -- and a single batch, and none
    monkeypatch.setenv("DPH_STREAM_DEPTH", "3")
    deep = list(mips.search_stream([qn, q, q, qn, q], q_texts=[list("abcdef")] * 5, **kw))
    assert [d == want for d in deep] == [False, True, True, False, True] and deep[0] == streamed[0] and deep[3] == streamed[0]
    assert list(mips.search_stream([q], q_texts=[list("abcdef")], **kw)) == [want] and list(mips.search_stream([], **kw)) == []
    for strat, sent in (("opt2", True), ("opt3", False), ("opt4", False)):
        kw2 = dict(top_k=5, aggregate=True, agg_strat=strat, return_sent=sent)
        assert list(mips.search_stream([q, q, q], q_texts=[list("abcdef")] * 3, **kw2)) == [mips.search(q, q_texts=list("abcdef"), **kw2)] * 3
