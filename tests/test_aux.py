"""Round 5: the filter on shards whose rows are NOT alike -- rogue dimensions (a mean code far from the other dimensions' for every row,
as in BERT-family vectors clipped by embed_utils.py:141-149), heavy-tailed row norms, runs of near-duplicates.  dph_index_finalize
gives such shards AUX ROWS (csrc/dph_scan.hip "the aux k-step"): a per-row norm bound and replica digits for the rogue dimensions.
CPU part: the layout agreement of the sharded path; `-m gpu`: the layout a shard chooses, the aux rows and digits against their
numpy restatement, the filter's definition, and parity with the oracle of faiss Index.search (index.py:200)."""
import numpy as np
import pytest

from oracle import mips_oracle as O


def test_agree_aux_layout_takes_the_widest_stride_the_first_replica_table_and_the_smallest_clamp():
    from densephrases_amd.dist import agree_aux_layout
    a = np.zeros((4, 28), np.int32)
    a[:, 3] = 64
    a[:, 4:] = -1
    a[1, :3] = [32, 8, 5]
    a[1, 4:9] = [729, 729, 77, 77, 381]
    a[2, :2] = [4, 4]
    a[2, 3] = 50
    a[3, :3] = [32, 8, 2]
    a[3, 4:6] = [5, 6]
    a[0, :3] = [16, 4, 3]
    a[0, 4:7] = [1, 2, 3]
    got = agree_aux_layout(a)
    assert got[:4].tolist() == [32, 8, 5, 50] and got[4:9].tolist() == [729, 729, 77, 77, 381] and (got[9:] == -1).all()
    assert agree_aux_layout(a[[0, 2]])[:7].tolist() == [16, 4, 3, 50, 1, 2, 3]
    a[0, :3] = 0
    a[0, 4:7] = -1
    b = a[[0, 2]]
    assert agree_aux_layout(b)[:4].tolist() == [4, 4, 0, 50] and (agree_aux_layout(b)[4:] == -1).all()
    assert agree_aux_layout(a[[0]])[:4].tolist() == [0, 0, 0, 64]


def host_digits(x, layout=None):
    """numpy replica of dph_quantize_kernel: q ~= sc * (128 * (q1 + replica digits) + q2).  Returns (q1 [n,768], q2 [n,768],
    X [n,768] = the sum of a dimension's replica digits, sc)."""
    x = x.astype(np.float32).astype(np.float64)
    reps = np.zeros(768, np.int64)
    q2max = 64
    if layout is not None:
        for d in layout[4:4 + int(layout[2])]:
            reps[int(d)] += 1
        q2max = int(layout[3])
    cap = 127.0 * (1 + reps)
    am = (np.abs(x) / (1 + reps)[None, :]).max(1)
    s = np.where(am > 0, am / 127.0, 1.0)
    u = x / s[:, None]
    Q1 = np.clip(np.rint(u), -cap[None, :], cap[None, :])
    q2 = np.clip(np.rint((u - Q1) * 128.0), -q2max, q2max)
    q1 = np.clip(Q1, -127, 127)
    return q1.astype(np.int64), q2.astype(np.int64), (Q1 - q1).astype(np.int64), s / 128.0


def host_aux_rows(xb, mu, layout, unit):
    """numpy replica of dph_aux_build_kernel"""
    stride, n_norm, n_rep = int(layout[0]), int(layout[1]), int(layout[2])
    d = xb.astype(np.int64) - mu[None, :].astype(np.int64)
    n2 = (d * d).sum(1)
    code = np.ceil(np.sqrt(n2.astype(np.float64)) / unit).astype(np.int64)
    code = np.where((code * unit) ** 2 < n2, code + 1, code)
    code = np.where(((code - 1) * unit) ** 2 >= n2, np.maximum(code - 1, 0), code)
    out = np.zeros((xb.shape[0], stride), np.int64)
    for s in range(n_norm):
        out[:, s] = np.clip(code - 127 * s, 0, 127)
    for s in range(n_rep):
        out[:, n_norm + s] = xb[:, int(layout[4 + s])]
    return out


gpu = pytest.mark.gpu


def _anisotropic_queries(rng, xb, n_q):
    from densephrases_amd.synth import ROGUE_DIMS, ROGUE_MEANS
    x = rng.normal(0, 0.5, (n_q, 768)).astype(np.float32)
    h = n_q // 2
    planted = rng.integers(0, xb.shape[0], h)
    x[:h] = xb[planted].astype(np.float32) / 20 - 2 + rng.normal(0, 0.1, (h, 768)).astype(np.float32)
    rm = (np.asarray(ROGUE_MEANS, np.float32) - 40.0) / 20.0
    x[h:, list(ROGUE_DIMS)] = rm[None, :] * (1.0 + rng.normal(0, 0.2, (n_q - h, len(rm)))).astype(np.float32)
    return x, planted


@gpu
@pytest.mark.parametrize("n_q,stride,force", [(128, 1, None), (200, 1, None), (96, 5, None), (128, 1, 32), (200, 3, 32)])
def test_aux_rows_digits_and_the_filter_definition_on_an_anisotropic_shard(n_q, stride, force):
    """The kind-4 dump: finalize finds its five rogue dimensions and gives the shard 16-byte aux rows (4 norm slots + 12 replica slots:
    what queries that look like rows need; the 32-byte layout with 8 + 24 slots is forced for the second half of the cases); rows and query digits equal
    their numpy restatement; under per-row bounds tau a visited row is emitted iff
    <q1, n> + sum_s aux[row][s] * qaux[q][s] > floor((tau - lmax) / 128), every emitted key carries the exact integer score
    128 * (<q1, n> + <replica digits, rogue codes>) + <q2, n>, and every row with I > tau is there."""
    from densephrases_amd import Shard
    from densephrases_amd.synth import ROGUE_DIMS
    from tests._devdata import device_rows
    n_rows = 70000
    xb = device_rows(n_rows, seed=42, kind=4)
    s = Shard(n_rows, device=0)
    s.upload(xb)
    s.finalize()
    lay = s.aux_layout()
    assert lay[:4].tolist() == [16, 4, 12, 64], lay
    if force:
        s.set_tuning("aux", force)
        lay = s.aux_layout()
        assert lay[:4].tolist() == [32, 8, 24, 64], lay
    n_norm, n_rep = int(lay[1]), int(lay[2])
    assert set(lay[4:4 + n_rep].tolist()) == set(ROGUE_DIMS) and (lay[4 + n_rep:] == -1).all()   # all five, at least one slot each
    mu = s.debug_mu()
    np.testing.assert_array_equal(mu, np.rint(xb.astype(np.float64).mean(0)).astype(np.int32))
    rng = np.random.default_rng(n_q)
    x, _ = _anisotropic_queries(rng, xb, n_q)
    q1, q2, X, sc = host_digits(x, lay)
    assert np.abs(X).max() > 127                                                       # the rogue digits really need replicas
    xi = xb.astype(np.int64)
    H = (q1 + X) @ xi.T                                                                # exact high-digit score incl. replicas
    ref = 128 * H + q2 @ xi.T
    assert np.abs(ref).max() < 2 ** 31
    visited = np.zeros(n_rows, bool)
    for t in range(0, (n_rows + 31) // 32, stride):
        visited[t * 32:(t + 1) * 32] = True
    tau = np.sort(ref[:, visited], axis=1)[:, -40].astype(np.int32)
    buckets, lost = s.debug_scan_buckets(x, tau=tau, tile_stride=stride)
    lmax = s.debug_lmax(n_q).astype(np.int64)
    aux, qaux, info = s.debug_aux(0, n_rows, n_q)
    assert info["stride"] == lay[0] and info["n_rep"] == n_rep and info["q2max"] == 64
    np.testing.assert_array_equal(aux.astype(np.int64), host_aux_rows(xb, mu, lay, info["unit"]))
    np.testing.assert_array_equal(lmax, q2 @ mu.astype(np.int64))                      # <q2, mu>: the norm part rides in the digits
    # the query side: replica digits sum to X per dimension, norm digits = ceil(unit * ||q2|| / 128)
    qa = qaux.astype(np.int64)
    assert (qa[:, n_norm + n_rep:] == 0).all() and (np.abs(qa) <= 127).all()
    for d in ROGUE_DIMS:
        slots = [n_norm + i for i in range(n_rep) if lay[4 + i] == d]
        np.testing.assert_array_equal(qa[:, slots].sum(1), X[:, d])
    bq = np.ceil(info["unit"] * np.sqrt((q2 * q2).sum(1).astype(np.float64)) / 128.0 * (1 + 1e-9)).astype(np.int64)
    for sl in range(n_norm):
        np.testing.assert_array_equal(qa[:, sl], bq)
    Hp = q1 @ xi.T + qa[:, :int(lay[0])] @ aux.astype(np.int64).T                      # what the 25 k-steps of the scan add up to
    thi = np.floor_divide(tau.astype(np.int64) - lmax, 128)
    for q in range(n_q):
        score, rows = buckets[q]
        r = rows.astype(np.int64)
        assert not lost[q] and len(set(r.tolist())) == r.size and visited[r].all()
        np.testing.assert_array_equal(score.astype(np.int64), ref[q, r])
        must = np.nonzero(visited & (Hp[q] > thi[q]))[0]
        assert np.array_equal(np.sort(r), must), f"q{q}: emitted set differs from the filter's definition"
        assert set(np.nonzero(visited & (ref[q] > tau[q]))[0].tolist()) <= set(must.tolist())
        assert must.size < 40 * 40                                                     # a per-row bound: the filter stays a filter


@gpu
def test_search_matches_the_oracle_on_a_million_anisotropic_rows():
    """parity with the restatement of faiss IndexFlatIP.search (index.py:200) on >= 1 M rows of the BERT-like
    dump -- planted queries, rogue-heavy random queries, batch of 256 (both scan kernels) -- and the first attempt certifies."""
    from densephrases_amd import Shard
    from tests._devdata import device_rows, gpu_flat_ip_search
    n_rows = 1_200_000
    xb = device_rows(n_rows, seed=7, kind=4, id_base=5000)
    s = Shard(n_rows, device=0, id_base=5000)
    s.fill_synthetic(seed=7, kind=4)
    s.finalize()
    assert s.aux_layout()[0] == 16
    rng = np.random.default_rng(11)
    for n_q in (64, 300):
        x, planted = _anisotropic_queries(rng, xb, n_q)
        D, I = s.search(x, 10)
        Dr, Ir, D64 = gpu_flat_ip_search(x, xb, 10, id_base=5000)
        ok, msg = O.topk_equivalent(D, I, D64, Ir)
        assert ok, msg
        st = s.stats()
        assert st["uncertified"] == 0 and st["certified_fast"] >= n_q - 2, st
    # ... and against the numpy oracle itself on the first rows (the torch restatement must not be the only witness)
    Dn, In, D64n = O.flat_ip_search(x[:6], xb, 10, id_base=5000)
    ok, msg = O.topk_equivalent(D[:6], I[:6], D64n, In)
    assert ok, msg


@gpu
def test_heavy_tailed_row_norms_get_norm_codes_only():
    """i.i.d. directions with log-normal row norms and no rogue dimension: 4-byte aux rows (per-row norm codes), the digits of rounds
    1-4 (no replicas), parity with the oracle, first-attempt certificates."""
    from densephrases_amd import Shard
    from tests._devdata import gpu_flat_ip_search
    rng = np.random.default_rng(3)
    n_rows = 400_000
    scale = np.exp(0.4 * rng.standard_normal(n_rows)).astype(np.float32)
    xb = O.float_to_int8(rng.standard_normal((n_rows, 768), dtype=np.float32) * (0.5 * scale)[:, None])
    s = Shard(n_rows, device=0)
    s.upload(xb)
    s.finalize()
    lay = s.aux_layout()
    assert lay[:4].tolist() == [4, 4, 0, 64], lay
    x = rng.normal(0, 0.5, (40, 768)).astype(np.float32)
    planted = rng.integers(0, n_rows, 20)
    x[:20] = xb[planted].astype(np.float32) / 20 - 2 + rng.normal(0, 0.1, (20, 768)).astype(np.float32)
    D, I = s.search(x, 10)
    Dr, Ir, D64 = gpu_flat_ip_search(x, xb, 10)
    ok, msg = O.topk_equivalent(D, I, D64, Ir)
    assert ok, msg
    st = s.stats()
    assert st["uncertified"] == 0 and st["certified_fast"] >= 38, st
    # the aux rows can be switched off (one shard-wide norm bound, as before round 5) and on again: same answer
    s.set_tuning("aux", 0)
    assert s.aux_layout()[0] == 0
    D0, I0 = s.search(x, 10)
    np.testing.assert_array_equal(I0, I)
    np.testing.assert_array_equal(D0, D)
    s.set_tuning("aux", -1)
    assert s.aux_layout()[0] == 4


@gpu
def test_two_shards_agree_on_one_aux_layout_and_merge_to_the_single_shard_answer():
    """Range shards of one dump may choose different layouts (here: the second shard is forced to none); the sharded path agrees on
    one (dist.agree_aux_layout) before integer scores cross a shard boundary, and the two-phase union-bound search then equals the
    single-shard answer."""
    import torch
    from densephrases_amd import Shard
    from densephrases_amd.dist import RecordLayout, ShardedSearcher, agree_aux_layout, exchange_and_merge, partition_rows
    from tests._devdata import device_rows
    n_rows, B, k, L, doc_len = 400_000, 16, 10, 10, 100
    xb = device_rows(n_rows, seed=9, kind=4)
    doc = (np.arange(n_rows) // doc_len).astype(np.int32)
    word = (np.arange(n_rows) % doc_len).astype(np.int32)
    doc_ids = np.arange(n_rows // doc_len, dtype=np.int32)
    f2o_off = np.arange(0, n_rows + 1, doc_len, dtype=np.int64)
    f2o = np.tile(np.arange(doc_len, dtype=np.int32), n_rows // doc_len)
    rng = np.random.default_rng(5)
    xs, _ = _anisotropic_queries(rng, xb, 2 * B)
    q = np.concatenate([xs[:B], xs[B:]], 1)
    dev = torch.device("cuda", 0)

    def make(lo, hi, aux=None):
        s = Shard(hi - lo, device=0, id_base=lo)
        s.upload(xb[lo:hi])
        s.set_idx2id(doc[lo:hi], word[lo:hi])
        s.set_f2o(doc_ids, f2o_off, f2o)
        s.finalize()
        if aux is not None:
            s.set_tuning("aux", aux)
        return s

    qd = torch.from_numpy(q).to(dev)
    full = ShardedSearcher(make(0, n_rows), B, k, L, device=dev)
    want = {kk: v.clone() for kk, v in full.step(qd).items()}
    assert int(want["status"].max()) == 0
    parts = partition_rows(n_rows, 2, align=doc_len)
    shards = [make(parts[0][0], parts[0][1]), make(parts[1][0], parts[1][1], aux=0)]
    lays = np.stack([s.aux_layout() for s in shards])
    assert lays[0, 0] == 16 and lays[1, 0] == 0
    agreed = agree_aux_layout(lays)
    for s in shards:
        s.set_aux_layout(agreed)
        np.testing.assert_array_equal(s.aux_layout(), agreed)
    ss = [ShardedSearcher(s, B, k, L, device=dev, union_bounds=True) for s in shards]
    top_all = torch.empty((2, 2 * B, 16), dtype=torch.int32, device=dev)
    for r, s in enumerate(ss):
        s.load_query(qd)
        s.sample()
        top_all[r].copy_(s.top)
    layout = RecordLayout(2 * B, k)
    rec_all = torch.zeros((2, layout.nbytes), dtype=torch.uint8, device=dev)
    for r, s in enumerate(ss):
        s.union_bound(top_all, 2)
        s.search_and_rescore()
        rec_all[r].copy_(s.rec)
    torch.cuda.synchronize()
    m = ss[0]
    m.world = 2

    class _NoDist:
        @staticmethod
        def all_gather_into_tensor(out, inp):
            pass

    D, I, best, pred, status = exchange_and_merge(layout, m.rec, rec_all, _NoDist, 2, m._merge)
    torch.cuda.synchronize()
    assert int(status.max()) == 0
    np.testing.assert_array_equal(I.cpu().numpy(), want["I"].cpu().numpy())
    np.testing.assert_array_equal(D.cpu().numpy(), want["D"].cpu().numpy())
    np.testing.assert_array_equal(pred.cpu().numpy(), want["pred"].cpu().numpy())


@gpu
@pytest.mark.parametrize("units", [0, 1])
def test_ivf_over_a_list_major_shard_of_anisotropic_rows(units):
    """The aux k-step in the OTHER instantiations of the scan: a list-major shard (masked scan with probe masks in flight next to the
    aux rows; unit scan over gathered query fragments) of the BERT-like dump -- aux rows follow the stored (permuted, padded) rows,
    padding rows keep a zero high-digit score -- against the IVF-flat oracle and, with every list probed, the flat oracle."""
    from densephrases_amd.ivf import train_centroids
    from tests._devdata import device_rows, gpu_flat_ip_search, gpu_ivf_flat_search
    from tests.test_ivf import _ivf_shard
    n_rows, nlist, nprobe, k = 120_000, 16, 4, 10
    xb = device_rows(n_rows, seed=13, kind=4)
    cent = train_centroids(xb[::6], nlist, iters=4, seed=3)
    s, assign = _ivf_shard(xb, cent, id_base=700, units=units)
    lay = s.aux_layout()
    assert lay[0] == 16 and lay[2] == 12, lay
    rng = np.random.default_rng(21)
    for n_q in (40, 300):
        x, _ = _anisotropic_queries(rng, xb, n_q)
        D, I = s.search_ivf(x, k, nprobe)
        Dr, Ir, D64 = gpu_ivf_flat_search(x, xb, cent, assign, nprobe, k)
        Ir = np.where(Ir >= 0, Ir + 700, -1)
        ok, msg = O.topk_equivalent(D, I, D64, Ir)
        assert ok, (n_q, msg)
        assert s.stats()["uncertified"] == 0
        Df, If = s.search(x, k)                                                        # exact search over the same list-major shard
        Drf, Irf, D64f = gpu_flat_ip_search(x, xb, k, id_base=700)
        ok, msg = O.topk_equivalent(Df, If, D64f, Irf)
        assert ok, (n_q, msg)
        assert s.stats()["uncertified"] == 0
