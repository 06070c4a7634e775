"""encoder.py's start/end-vector scoring on libdph against a plain PyTorch fp32 reference of the
same contractions (/root/reference/densephrases/encoder.py:206-208, 383-386)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,M", [(64, 20), (3, 1), (1, 400), (7, 33)])
def test_phrase_logits_match_torch_and_backprop_into_the_query(B, M):
    import torch
    from densephrases_amd.scoring import phrase_logits
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + M)
    dev = torch.device("cuda", 0)
    qs = torch.randn(B, 1, 768, generator=g).to(dev).requires_grad_(True)
    qe = torch.randn(B, 1, 768, generator=g).to(dev).requires_grad_(True)
    sv = (torch.randn(B, M, 768, generator=g) * 0.6).to(dev)
    ev = (torch.randn(B, M, 768, generator=g) * 0.6).to(dev)
    s, e, lg = phrase_logits(qs, qe, sv, ev)
    # the reference's three lines, verbatim
    s_ref = qs.matmul(sv.transpose(1, 2)).squeeze(1)
    e_ref = qe.matmul(ev.transpose(1, 2)).squeeze(1)
    lg_ref = s_ref + e_ref
    tol = dict(rtol=1e-5, atol=2e-4)          # fp32, different summation order over 768 terms of magnitude ~0.6
    torch.testing.assert_close(s, s_ref, **tol)
    torch.testing.assert_close(e, e_ref, **tol)
    torch.testing.assert_close(lg, lg_ref, **tol)
    w = torch.randn(B, M, generator=g).to(dev)
    (lg * w).sum().backward()
    gs, ge = qs.grad.clone(), qe.grad.clone()
    qs.grad = None
    qe.grad = None
    (lg_ref * w).sum().backward()
    torch.testing.assert_close(gs, qs.grad, rtol=1e-5, atol=2e-4)
    torch.testing.assert_close(ge, qe.grad, rtol=1e-5, atol=2e-4)


def test_dense_logits_match_torch():
    import torch
    from densephrases_amd.scoring import dense_logits
    g = torch.Generator(device="cpu").manual_seed(5)
    dev = torch.device("cuda", 0)
    bs, T = 6, 37
    start = torch.randn(bs, T, 768, generator=g).to(dev)
    end = torch.randn(bs, T, 768, generator=g).to(dev)
    qs = torch.randn(bs, 1, 768, generator=g).to(dev)
    qe = torch.randn(bs, 1, 768, generator=g).to(dev)
    s, e, d = dense_logits(start, end, qs, qe)
    s_ref = start.matmul(qs.transpose(1, 2)).squeeze(-1)
    e_ref = end.matmul(qe.transpose(1, 2)).squeeze(-1)
    d_ref = s_ref.unsqueeze(2) + e_ref.unsqueeze(1)
    torch.testing.assert_close(s, s_ref, rtol=1e-5, atol=5e-4)
    torch.testing.assert_close(e, e_ref, rtol=1e-5, atol=5e-4)
    torch.testing.assert_close(d, d_ref, rtol=1e-5, atol=1e-3)
    assert torch.equal(d, s.unsqueeze(2) + e.unsqueeze(1))       # the add itself is exact


def test_dense_logits_backprop_into_the_phrase_vectors_and_the_query():
    """Encoder.forward is training code (encoder.py:206-208: start / end come from the phrase encoder, dense_logits feeds
    the loss): the gradients w.r.t. start, end AND the query must be those of the three torch lines"""
    import torch
    from densephrases_amd.scoring import dense_logits
    g = torch.Generator(device="cpu").manual_seed(9)
    dev = torch.device("cuda", 0)
    bs, T = 3, 11
    leaves = [torch.randn(bs, T, 768, generator=g).to(dev).requires_grad_(True), torch.randn(bs, T, 768, generator=g).to(dev).requires_grad_(True),
              torch.randn(bs, 1, 768, generator=g).to(dev).requires_grad_(True), torch.randn(bs, 1, 768, generator=g).to(dev).requires_grad_(True)]
    start, end, qs, qe = leaves
    w = torch.randn(bs, T, T, generator=g).to(dev)
    _, _, d = dense_logits(start, end, qs, qe)
    assert d.requires_grad
    (d * w).sum().backward()
    got = [t.grad.clone() for t in leaves]
    for t in leaves:
        t.grad = None
    s_ref = start.matmul(qs.transpose(1, 2)).squeeze(-1)
    e_ref = end.matmul(qe.transpose(1, 2)).squeeze(-1)
    ((s_ref.unsqueeze(2) + e_ref.unsqueeze(1)) * w).sum().backward()
    for a, t in zip(got, leaves):
        torch.testing.assert_close(a, t.grad, rtol=1e-5, atol=1e-3)


def test_scoring_the_vectors_search_returns(tmp_path):
    """End to end with the index: MIPS.search(return_idxs=True) -> [B, 2k, 768] vectors -> phrase_logits reproduces the
    first-stage score sum of every returned phrase (start <q_s, v_s> + end <q_e, v_e>)."""
    import torch
    from densephrases_amd import DocMeta, DocStore, MIPS
    from densephrases_amd.scoring import phrase_logits
    from oracle.synth_dump import make_dump, make_queries
    docs = make_dump(seed=3, n_docs=8, d=768)
    mips = MIPS.from_store(DocStore([DocMeta(m.doc_idx, m.title, m.context, m.f2o_start, m.word2char_start, m.word2char_end,
                                             m.start) for m in docs]))
    q = make_queries(np.random.default_rng(1), mips.store.rows, 5)
    outs = mips.search(q, top_k=4, return_idxs=True)
    M = min(len(o) for o in outs)
    sv = torch.tensor(np.stack([[r["start_vec"] for r in o[:M]] for o in outs])).cuda()
    ev = torch.tensor(np.stack([[r["end_vec"] for r in o[:M]] for o in outs])).cuda()
    qt = torch.from_numpy(q).cuda()
    _, _, lg = phrase_logits(qt[:, None, :768], qt[:, None, 768:], sv, ev)
    want = np.array([[r["score"] for r in o[:M]] for o in outs])
    np.testing.assert_allclose(lg.cpu().numpy(), want, rtol=1e-5, atol=1e-3)
