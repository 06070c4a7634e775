"""Two batches in flight on one shard (include/dph.h "two batches in flight"; densephrases_amd.dist.PipelinedSearcher): twins of an
index, CU-masked streams, the search in two stages.  The launches of a batch are those of the plain step, so everything is compared
bit for bit with ShardedSearcher.step on the same shard -- which the other GPU tests hold against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ("D", "I", "best", "pred", "status")


def _shard(n_rows, seed, dup=0):
    from densephrases_amd import Shard
    rng = np.random.default_rng(seed)
    xb = np.clip(np.rint(40 + 12 * rng.standard_normal((n_rows, 768), dtype=np.float32)), -128, 127).astype(np.int8)
    if dup:
        xb[rng.choice(np.arange(1000, n_rows), dup, replace=False)] = xb[777]
    s = Shard(n_rows, device=0)
    s.upload(xb)
    s.set_idx2id((np.arange(n_rows) // 100).astype(np.int32), (np.arange(n_rows) % 100).astype(np.int32))
    s.set_f2o(np.arange(n_rows // 100, dtype=np.int32), np.arange(0, n_rows + 1, 100, dtype=np.int64),
              np.tile(np.arange(100, dtype=np.int32), n_rows // 100))
    s.finalize()
    return s, xb, rng


def _batches(rng, xb, n_batches, B, hot=None):
    qs = []
    for i in range(n_batches):
        pick = rng.integers(0, xb.shape[0], 2 * B)
        x = (xb[pick].astype(np.float32) / 20 - 2 + rng.normal(0, 0.3, (2 * B, 768))).astype(np.float32)
        q = np.concatenate([x[:B], x[B:]], 1)
        if hot is not None and i == 2:
            q[0, :768] = hot.astype(np.float32) / 20 - 2          # a row with thousands of exact copies of its best match: retry chain + fp64 scan
        qs.append(q)
    return qs


@pytest.mark.parametrize("n_rows,B,dup", [(6000, 3, 0), (400_000, 4, 3000), (1_200_000, 64, 0), (1_200_000, 100, 0)])
def test_two_batches_in_flight_return_what_the_plain_step_returns(n_rows, B, dup):
    """(6000 rows: the shard is scanned cold, no sampled level; 400 k with 3000 copies of one row: batch 2 goes through the retry
    chain and the fp64 scan inside a lane; 1.2 M rows: two sampled levels, the full scan fused with the finer one, 128 rows (qb 1) and 200 rows (qb 2).)"""
    import torch
    from densephrases_amd.dist import PipelinedSearcher, ShardedSearcher
    k, L = 10, 5
    s, xb, rng = _shard(n_rows, 5, dup)
    qs = [torch.from_numpy(q).cuda() for q in _batches(rng, xb, 6, B, hot=xb[777] if dup else None)]
    plain = ShardedSearcher(s, B, k, L, device=torch.device("cuda", 0))
    want = []
    for q in qs:
        out = plain.step(q)
        want.append({key: out[key].cpu().numpy().copy() for key in KEYS})
    if dup:
        assert s.stats()["rows"] == 2 * B
    pipe = PipelinedSearcher(s, B, k, L, side_cus=8)
    got = []
    for q in qs:
        out = pipe.step(q)
        if out is not None:
            got.append({key: out[key].cpu().numpy().copy() for key in KEYS})
    out = pipe.flush()
    got.append({key: out[key].cpu().numpy().copy() for key in KEYS})
    assert pipe.flush() is None and len(got) == len(want)
    for t, (g, w) in enumerate(zip(got, want)):
        for key in KEYS:
            np.testing.assert_array_equal(g[key], w[key], err_msg=f"batch {t}: {key}")
        assert (g["status"] == 0).all()
    if dup:
        st = pipe.lanes[0].shard.stats()            # batch 2 ran on lane 0
        assert st["uncertified"] == 0
    # the shard itself still answers (its own scratch was never lent out)
    out = plain.step(qs[0])
    np.testing.assert_array_equal(out["I"].cpu().numpy(), want[0]["I"])
    pipe.close()
    s.close()


def test_a_twin_shares_the_rows_and_refuses_to_change_them():
    from densephrases_amd import Shard
    from densephrases_amd._lib import DphError
    s, xb, rng = _shard(50_000, 9)
    x = (xb[:6].astype(np.float32) / 20 - 2).astype(np.float32)
    D0, I0 = s.search(x, 10)
    t = s.twin()
    D1, I1 = t.search(x, 10)
    np.testing.assert_array_equal(I0, I1)
    np.testing.assert_array_equal(D0, D1)
    np.testing.assert_array_equal(t.reconstruct(123), s.reconstruct(123))
    d0, w0 = s.id2docword(I0)
    d1, w1 = t.id2docword(I0)
    np.testing.assert_array_equal(d0, d1)
    np.testing.assert_array_equal(w0, w1)
    for h in (s, t):                                   # rows and metadata are frozen while twins exist
        with pytest.raises(DphError):
            h.upload(xb[:10], 0)
        with pytest.raises(DphError):
            h.finalize()
        with pytest.raises(DphError):
            h.set_codec(-2.0, 20.0)
    with pytest.raises(DphError):
        t.twin()                                       # twins of the index, not of a twin
    t.close()
    s.finalize()                                       # no twin left: the index is its own again
    t2 = s.twin()
    s.close()                                          # closes the twin first
    assert not t2._h


def test_cu_range_streams_partition_the_chip():
    """A kernel of 248 persistent workgroups on the 248-CU stream and the same search on an unrestricted stream agree; a range beyond the
    device is refused."""
    import torch
    from densephrases_amd import _lib
    from densephrases_amd._lib import DphError
    cus = int(torch.cuda.get_device_properties(0).multi_processor_count)
    with pytest.raises(DphError):
        _lib.stream_create_cu_range(0, cus - 4, 8)
    raw = _lib.stream_create_cu_range(0, 0, 4)          # a stream torch never sees can be destroyed again
    _lib.stream_destroy(raw)
    st = torch.cuda.ExternalStream(_lib.cu_range_stream(0, 8, cus - 8), device=torch.device("cuda", 0))
    s, xb, rng = _shard(300_000, 11)
    t = s.twin()
    t.set_tuning("scan_grid", cus - 8)
    x = torch.from_numpy((xb[:40].astype(np.float32) / 20 - 2).astype(np.float32)).cuda()
    D = torch.empty((40, 10), dtype=torch.float32, device="cuda")
    I = torch.empty((40, 10), dtype=torch.int64, device="cuda")
    status = torch.empty((40,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    t.search_dev(x.data_ptr(), 40, 10, D.data_ptr(), I.data_ptr(), status.data_ptr(), st.cuda_stream)
    st.synchronize()
    D0, I0 = s.search(x.cpu().numpy(), 10)
    np.testing.assert_array_equal(I.cpu().numpy(), I0)
    np.testing.assert_array_equal(D.cpu().numpy(), D0)
    assert int(status.abs().sum()) == 0
    with pytest.raises(DphError):
        t.set_tuning("scan_grid", cus)                # scratch is sized by it: before the first search only
    s.close()
