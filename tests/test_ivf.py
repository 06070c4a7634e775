"""IVF (BASELINE.json configs[3]): list builder invariants on the CPU, libdph's list-major shard against the oracle's
IVF-flat restatement on the GPU."""
import numpy as np
import pytest

from oracle import mips_oracle as O


def _clustered_db(rng, n, n_clusters=32, d=768):
    centres = rng.normal(0.0, 0.5, (n_clusters, d)).astype(np.float32)
    which = rng.integers(0, n_clusters, n)
    x = centres[which] + rng.normal(0.0, 0.25, (n, d)).astype(np.float32)
    return O.float_to_int8(x), centres


def test_list_major_builder_invariants():
    from densephrases_amd.ivf import assign_lists, build_list_major, train_centroids
    rng = np.random.default_rng(0)
    xb, _ = _clustered_db(rng, 3000, 8)
    cent = train_centroids(xb, 16, iters=4, seed=1)
    assert cent.shape == (16, 768) and np.isfinite(cent).all()
    a = assign_lists(xb, cent)
    want = np.argmax((xb.astype(np.float32) / 20 - 2).astype(np.float64) @ cent.astype(np.float64).T, 1)
    np.testing.assert_array_equal(a, want)
    stored, row_ids, tile_list = build_list_major(xb, a, 16, id_base=100)
    assert stored.shape[0] % 32 == 0 and tile_list.shape[0] == stored.shape[0] // 32
    real = row_ids >= 0
    assert sorted((row_ids[real] - 100).tolist()) == list(range(3000))          # a permutation
    np.testing.assert_array_equal(stored[real], xb[row_ids[real] - 100])
    assert (stored[~real] == 0).all()
    lists_of_rows = np.repeat(tile_list, 32)
    np.testing.assert_array_equal(lists_of_rows[real], a[row_ids[real] - 100])  # every tile holds one list
    for l in range(16):                                                         # id order inside a list
        ids = row_ids[(lists_of_rows == l) & real]
        assert (np.diff(ids) > 0).all()


def test_oracle_ivf_with_all_lists_probed_is_flat():
    rng = np.random.default_rng(1)
    xb, _ = _clustered_db(rng, 500, 4, d=32)
    cent = rng.normal(0, 1, (6, 32)).astype(np.float32)
    assign = np.argmax((xb.astype(np.float32) / 20 - 2).astype(np.float64) @ cent.astype(np.float64).T, 1)
    q = rng.normal(0, 1, (5, 32)).astype(np.float32)
    D1, I1, _ = O.ivf_flat_search(q, xb, cent, assign, 6, 7)
    D2, I2, _ = O.flat_ip_search(q, xb, 7)
    np.testing.assert_array_equal(I1, I2)
    np.testing.assert_array_equal(D1, D2)


def _ivf_shard(xb, cent, id_base=0, units=-1):
    """units: the "ivf_units" tuning key -- 1 = unit scan (work queue of (list chunk, segment) units, up to 1024 query
    rows per pass), 0 = masked scan (every tile, probe mask per tile), -1 = the library's choice"""
    from densephrases_amd import Shard
    from densephrases_amd.ivf import assign_lists, build_list_major
    a = assign_lists(xb, cent)
    stored, row_ids, tile_list = build_list_major(xb, a, cent.shape[0], id_base=id_base)
    s = Shard(stored.shape[0], device=0, id_base=id_base)
    s.upload(stored)
    s.set_row_ids(row_ids, xb.shape[0])
    s.set_ivf(cent, tile_list)
    s.finalize()
    s.set_tuning("ivf_units", units)
    return s, a


@pytest.mark.gpu
@pytest.mark.parametrize("units", [0, 1])
@pytest.mark.parametrize("n_rows,nlist,nprobe,n_q,k", [(30000, 64, 8, 130, 10), (5000, 16, 1, 7, 5), (3000, 8, 8, 3, 20),
                                                       (30000, 8, 3, 700, 10)])
def test_ivf_search_matches_oracle(n_rows, nlist, nprobe, n_q, k, units):
    """(30000, 8, 3, 700): lists of ~120 tiles probed by ~260 query rows each -- several 128-slot chunks per list, a
    cold ladder level, one pass of 700 rows in the unit scan (three passes in the masked scan)."""
    from densephrases_amd.ivf import train_centroids
    rng = np.random.default_rng(n_rows + nlist)
    xb, centres = _clustered_db(rng, n_rows, 24)
    cent = train_centroids(xb, nlist, iters=5, seed=3)
    s, assign = _ivf_shard(xb, cent, id_base=500, units=units)
    x = (centres[rng.integers(0, 24, n_q)] + rng.normal(0, 0.3, (n_q, 768))).astype(np.float32)
    D, I = s.search_ivf(x, k, nprobe)
    from tests._devdata import gpu_flat_ip_search, gpu_ivf_flat_search
    if n_q > 200:
        # the numpy oracle walks the query rows one by one (40 s for 700 of them): its torch restatement here, and the numpy oracle
        # itself on a slice of the rows as the witness of that restatement
        Dr, Ir, D64 = gpu_ivf_flat_search(x, xb, cent, assign, nprobe, k)
        Dn, In, D64n = O.ivf_flat_search(x[:40], xb, cent, assign, nprobe, k)
        np.testing.assert_array_equal(Ir[:40], In)
        np.testing.assert_allclose(D64[:40], D64n, rtol=1e-13, atol=1e-9)
    else:
        Dr, Ir, D64 = O.ivf_flat_search(x, xb, cent, assign, nprobe, k)
    Ir = np.where(Ir >= 0, Ir + 500, -1)
    ok, msg = O.topk_equivalent(D, I, D64, Ir)
    assert ok, msg
    assert s.stats()["uncertified"] == 0
    # the exact search over the same list-major shard is the flat oracle
    Df, If = s.search(x, k)
    Drf, Irf, D64f = (gpu_flat_ip_search if n_q > 200 else O.flat_ip_search)(x, xb, k, id_base=500)
    ok, msg = O.topk_equivalent(Df, If, D64f, Irf)
    assert ok, msg
    assert s.ntotal == n_rows
    np.testing.assert_array_equal(s.reconstruct(500 + 17), O.int8_to_float(xb[17]))


@pytest.mark.gpu
def test_ivf_recall_and_window_on_list_major_shard():
    """recall@k of IVF against exact search on clustered data (BASELINE config 4 asks for +-0.1 at nprobe/nlist = 1/16;
    here 8/64), and the window re-score addressed by id on a permuted shard equals the flat shard's."""
    from densephrases_amd import Shard
    from densephrases_amd.ivf import train_centroids
    rng = np.random.default_rng(77)
    n_rows = 40000
    xb, centres = _clustered_db(rng, n_rows, 48)
    cent = train_centroids(xb, 64, iters=6, seed=5)
    s, _ = _ivf_shard(xb, cent)
    x = (xb[rng.integers(0, n_rows, 64)].astype(np.float32) / 20 - 2 + rng.normal(0, 0.1, (64, 768))).astype(np.float32)
    Di, Ii = s.search_ivf(x, 5, 8)
    Df, If = s.search(x, 5)
    r1 = float((Ii[:, 0] == If[:, 0]).mean())
    r5 = float(np.mean([len(set(a) & set(b)) / 5.0 for a, b in zip(Ii, If)]))
    assert r1 >= 0.9 and r5 >= 0.9, (r1, r5)
    # window re-score through the id -> row map
    doc = (np.arange(n_rows) // 50).astype(np.int32)
    word = (np.arange(n_rows) % 50).astype(np.int32)
    did = np.arange(n_rows // 50, dtype=np.int32)
    off = np.arange(0, n_rows + 1, 50, dtype=np.int64)
    f2o = np.tile(np.arange(50, dtype=np.int32), n_rows // 50)
    flat = Shard(n_rows, device=0)
    flat.upload(xb)
    for sh in (s, flat):
        sh.set_idx2id(doc, word)
        sh.set_f2o(did, off, f2o)
    flat.finalize()
    ids = If[:8].copy()
    d_, w_ = flat.id2docword(ids)
    d2, w2 = s.id2docword(ids)
    np.testing.assert_array_equal(d_, d2)
    np.testing.assert_array_equal(w_, w2)
    first = Df[:8]
    for direction in (0, 1):
        a = flat.rescore(direction, x[:8], 5, 10, ids, d_, w_, first, want_vecs=True)
        b = s.rescore(direction, x[:8], 5, 10, ids, d_, w_, first, want_vecs=True)
        for u, v in zip(a, b):
            np.testing.assert_array_equal(u, v)


@pytest.mark.gpu
@pytest.mark.parametrize("units", [0, 1])
def test_ivf_coarse_quantizer_with_many_lists(units):
    """nlist = 20000 (beyond what the round-1 LDS-resident coarse kernel could hold): scores from the MFMA GEMM, radix
    select of the nprobe-th, float64 re-rank of the band -- the probed sets, hence the results, equal the float64
    oracle's."""
    rng = np.random.default_rng(31)
    n_rows, nlist, nprobe, n_q, k = 40000, 20000, 300, 40, 10
    xb, centres = _clustered_db(rng, n_rows, 24)
    cent = O.int8_to_float(xb[rng.choice(n_rows, nlist, replace=False)]).astype(np.float32)
    cent[5] = cent[4]                                  # duplicate centroids: exact score ties, resolved by list id
    s, assign = _ivf_shard(xb, cent, units=units)      # units = 1: ~12000 one-tile units in the work queue
    x = (centres[rng.integers(0, 24, n_q)] + rng.normal(0, 0.3, (n_q, 768))).astype(np.float32)
    D, I = s.search_ivf(x, k, nprobe)
    Dr, Ir, D64 = O.ivf_flat_search(x, xb, cent, assign, nprobe, k)
    ok, msg = O.topk_equivalent(D, I, D64, Ir)
    assert ok, msg
    assert s.stats()["uncertified"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("units", [0, 1])
def test_ivf_4096_lists_nprobe_256_recall_on_the_mixture(units):
    """BASELINE.json configs[3]: IVF-4096, nprobe 256, batches of 256 queries, on a mixture of 4096 Gaussians
    (sigma_between 0.5, sigma_within 0.25): recall@1 / @5 of the IVF search against the exact search within
    0.1 of 1.0, k-means + list assignment on the GPU (libdph's MFMA GEMM), and the GPU assignment equals the float64
    host assignment row for row."""
    from densephrases_amd.ivf import assign_lists, assign_lists_gpu, build_list_major, train_centroids
    from densephrases_amd import Shard
    rng = np.random.default_rng(404)
    n_rows, nlist, nprobe, B = 300000, 4096, 256, 256
    centres = rng.normal(0, 0.5, (nlist, 768)).astype(np.float32)
    a0 = rng.integers(0, nlist, n_rows)
    xb = np.empty((n_rows, 768), np.int8)
    for r0 in range(0, n_rows, 50000):
        xb[r0:r0 + 50000] = O.float_to_int8(centres[a0[r0:r0 + 50000]] + rng.normal(0, 0.25, (min(50000, n_rows - r0), 768)).astype(np.float32))
    cent = train_centroids(xb[:150000], nlist, iters=4, seed=1)
    assign = assign_lists_gpu(xb, cent)
    np.testing.assert_array_equal(assign[:20000], assign_lists(xb[:20000], cent))
    stored, row_ids, tile_list = build_list_major(xb, assign, nlist)
    s = Shard(stored.shape[0], device=0)
    s.upload(stored)
    s.set_row_ids(row_ids, n_rows)
    s.set_ivf(cent, tile_list)
    s.finalize()
    s.set_tuning("ivf_units", units)
    pick = rng.integers(0, n_rows, 2 * B)
    x = (O.int8_to_float(xb[pick]) + rng.normal(0, 0.3, (2 * B, 768))).astype(np.float32)      # 512 query rows: two passes of 256
    Di, Ii = s.search_ivf(x, 5, nprobe)
    assert s.stats()["uncertified"] == 0
    Df, If = s.search(x, 5)
    r1 = float((Ii[:, 0] == If[:, 0]).mean())
    r5 = float(np.mean([len(set(a) & set(b)) / 5.0 for a, b in zip(Ii, If)]))
    assert r1 >= 0.9 and r5 >= 0.9, (r1, r5)
    # and against the float64 oracle on a slice of the batch (exact in-list scores over exactly the probed lists)
    Dr, Ir, D64 = O.ivf_flat_search(x[:24], xb, cent, assign, nprobe, 5)
    ok, msg = O.topk_equivalent(Di[:24], Ii[:24], D64, Ir)
    assert ok, msg


@pytest.mark.gpu
@pytest.mark.parametrize("units", [0, 1])
def test_ivf_on_two_list_major_shards_equals_one_shard(units):
    """configs[3] is a multi-GPU config: each shard holds ITS rows of every list (list-major inside the shard) and a
    replica of the centroids; with the tuning key "nprobe" the sharded two-phase search (sample -> union bound ->
    bounded search -> merge) probes the same lists on every shard and returns what the single IVF shard returns."""
    import torch
    from densephrases_amd import _lib, Shard
    from densephrases_amd.ivf import assign_lists, build_list_major, train_centroids
    rng = np.random.default_rng(9)
    n_rows, nlist, nprobe, n_q, k = 60000, 64, 8, 48, 10
    xb, centres = _clustered_db(rng, n_rows, 40)
    cent = train_centroids(xb, nlist, iters=4, seed=2)
    assign = assign_lists(xb, cent)
    x = (centres[rng.integers(0, 40, n_q)] + rng.normal(0, 0.3, (n_q, 768))).astype(np.float32)

    def make(lo, hi):
        stored, row_ids, tile_list = build_list_major(xb[lo:hi], assign[lo:hi], nlist, id_base=lo)
        s = Shard(stored.shape[0], device=0, id_base=lo)
        s.upload(stored)
        s.set_row_ids(row_ids, hi - lo)
        s.set_ivf(cent, tile_list)
        s.finalize()
        s.set_tuning("nprobe", nprobe)
        s.set_tuning("ivf_units", units)
        return s

    one = make(0, n_rows)
    want_D, want_I = one.search(x, k)                      # "nprobe" tuning: the plain entry point searches IVF
    ref_D, ref_I = one.search_ivf(x, k, nprobe)
    np.testing.assert_array_equal(want_I, ref_I)
    dev = torch.device("cuda", 0)
    xd = torch.from_numpy(x).to(dev)
    cut = 31000
    shards = [make(0, cut), make(cut, n_rows)]
    tops = torch.empty((2, n_q, 16), dtype=torch.int32, device=dev)
    for r, s in enumerate(shards):
        s.search_sample_dev(xd.data_ptr(), n_q, tops[r].data_ptr())
    tau = torch.empty(n_q, dtype=torch.int32, device=dev)
    _lib.union_bounds_dev(0, tops.data_ptr(), 2, n_q, tau.data_ptr())
    D = torch.empty((2, n_q, k), dtype=torch.float32, device=dev)
    I = torch.empty((2, n_q, k), dtype=torch.int64, device=dev)
    st = torch.empty((2, n_q), dtype=torch.int32, device=dev)
    bd = torch.empty((2, n_q), dtype=torch.float64, device=dev)
    for r, s in enumerate(shards):
        s.search_bounded_dev(xd.data_ptr(), n_q, k, tau.data_ptr(), D[r].data_ptr(), I[r].data_ptr(), st[r].data_ptr(), bd[r].data_ptr())
    Dm = torch.empty((n_q, k), dtype=torch.float32, device=dev)
    Im = torch.empty((n_q, k), dtype=torch.int64, device=dev)
    _lib.merge_topk_dev(0, D.data_ptr(), I.data_ptr(), 2, n_q, k, Dm.data_ptr(), Im.data_ptr())
    torch.cuda.synchronize()
    np.testing.assert_array_equal(Im.cpu().numpy(), want_I)
    np.testing.assert_array_equal(Dm.cpu().numpy(), want_D)


@pytest.mark.gpu
def test_unit_scan_equals_masked_scan_on_long_lists():
    """2 M rows generated on the device, 24 inverted lists of ~2600 tiles (ten 256-tile segments each), batches of 40
    and 900 query rows: the unit scan (ladder: cold level on tile 0 of every probed list, a stride-32 level inside the
    lists, then the full units) and the masked scan return the same ids and scores, everything certified, and a
    slice agrees with the float64 oracle over exactly the probed lists."""
    import torch
    from densephrases_amd import Shard
    from densephrases_amd.synth import synthetic_rows
    n, nlist, nprobe, k = 2_000_000 // 32 * 32, 24, 5, 10
    rng = np.random.default_rng(5)
    s = Shard(n, device=0)
    s.fill_synthetic(seed=11)
    s.set_row_ids(np.arange(n, dtype=np.int64), n)
    cuts = np.sort(rng.choice(np.arange(1, n // 32), nlist - 1, replace=False))
    tile_list = np.zeros(n // 32, dtype=np.int32)
    tile_list[cuts] = 1
    tile_list = np.cumsum(tile_list).astype(np.int32)
    cent = rng.normal(0, 0.5, (nlist, 768)).astype(np.float32)
    s.set_ivf(cent, tile_list)
    s.finalize()
    for n_q in (40, 900):
        x = rng.normal(0, 0.5, (n_q, 768)).astype(np.float32)
        # plant a near-duplicate of a stored row in every fourth query: a clear winner when its list is probed
        rows = rng.integers(0, n, n_q)
        for i in range(0, n_q, 4):
            x[i] = O.int8_to_float(synthetic_rows(int(rows[i]), 1, 11)[0]) + rng.normal(0, 0.05, 768)
        out = {}
        for units in (1, 0):
            s.set_tuning("ivf_units", units)
            out[units] = s.search_ivf(x, k, nprobe)
            assert s.stats()["uncertified"] == 0
        np.testing.assert_array_equal(out[1][1], out[0][1])
        np.testing.assert_array_equal(out[1][0], out[0][0])
    # independent float64 brute force (plain torch over the resident rows, no libdph kernel) over exactly the probed
    # lists, for a few rows of the last batch
    class _Rows:
        def __init__(self, ptr):
            self.__cuda_array_interface__ = {"shape": (n, 768), "typestr": "|i1", "data": (int(ptr), False), "version": 2}
    dev = torch.device("cuda", 0)
    db = torch.as_tensor(_Rows(s.rows_dev_ptr()), device=dev)
    x8 = x[:8]
    probe = np.argsort(-(x8.astype(np.float64) @ cent.astype(np.float64).T), axis=1, kind="stable")[:, :nprobe]
    starts = np.concatenate([[0], cuts, [n // 32]]) * 32
    for qi in range(8):
        q = torch.from_numpy(x8[qi]).to(dev).to(torch.float64)
        sc, ids = [], []
        for l in probe[qi]:
            lo, hi = int(starts[l]), int(starts[l + 1])
            v = (db[lo:hi].to(torch.float32) / 20.0 - 2.0).to(torch.float64) @ q
            t = torch.topk(v, min(k, hi - lo))
            sc.append(t.values.cpu().numpy())
            ids.append(t.indices.cpu().numpy() + lo)
        sc, ids = np.concatenate(sc), np.concatenate(ids)
        order = np.lexsort((ids, -sc))[:k]
        np.testing.assert_array_equal(out[1][1][qi], ids[order])


@pytest.mark.gpu
def test_mips_class_with_ivf_lists():
    """The product MIPS class over a list-major shard (MIPS(ivf=...)): with every list probed it returns the reference's
    golden results (index.py run unmodified, tests/golden) dict for dict; with `nprobe` < nlist, `search_dense` returns
    what the oracle's IVF restatement returns over the same centroids and assignment -- and `nprobe` is the argument of
    `search` / `search_dense`, as in the reference (index.py:189,424)."""
    from densephrases_amd import DocMeta, DocStore, MIPS
    from tests._golden import compare_results, load_cases, load_toy_docs
    cases, vecs = load_cases()
    docs = load_toy_docs()
    store = DocStore([DocMeta(m.doc_idx, m.title, m.context, m.f2o_start, m.word2char_start, m.word2char_end, m.start)
                      for m in docs])
    nlist = 6
    mips = MIPS.from_store(store, ivf={"nlist": nlist, "nprobe": nlist, "iters": 4, "seed": 3})
    assert mips.index.ntotal == store.n_rows
    done = 0
    for c in cases:
        if c["return_idxs"] and c["branch"] == "hdf5":
            continue
        got = mips.search(c["query_arr"].astype(np.float64), q_texts=[f"q{i}" for i in range(c["B"])], nprobe=nlist,
                          top_k=c["top_k"], aggregate=c["aggregate"], return_idxs=c["return_idxs"],
                          max_answer_length=c["L"], agg_strat=c["agg_strat"], return_sent=c["return_sent"])
        compare_results(got, c["results"], vecs)
        done += 1
        if done == 6:
            break
    assert done >= 3
    # nprobe < nlist through the reference's own argument
    c = cases[0]
    q = c["query_arr"].astype(np.float32)
    k = c["top_k"]
    sd, sw, sI, ed, ew, eI, sS, eS = mips.search_dense(q, nprobe=2, top_k=k)
    assert mips.ivf["nprobe"] == nlist          # the argument holds for the call, the configured nprobe comes back
    stacked = np.concatenate(np.split(q, 2, axis=1), axis=0)
    Dr, Ir, D64 = O.ivf_flat_search(stacked, store.rows, mips.ivf["centroids"], mips.ivf["assign"], 2, k)
    ok, msg = O.topk_equivalent(np.concatenate([sS, eS]), np.concatenate([sI, eI]), D64, Ir)
    assert ok, msg
    # the entry points without the reference's nprobe argument search under the CONFIGURED nprobe before and after a
    # search(nprobe=...) call (round-2 advice: the first search() used to overwrite it for good), and take it per call too
    a = mips.search_device(q, top_k=k)
    mips.search(q.astype(np.float64), nprobe=2, top_k=k)
    b = mips.search_device(q, top_k=k)
    full = mips.search(q.astype(np.float64), nprobe=nlist, top_k=k)
    two = mips.search(q.astype(np.float64), nprobe=2, top_k=k)
    key = lambda res: [[(r["doc_idx"], r["start_idx"], r["end_idx"]) for r in per_q] for per_q in res]      # noqa: E731
    assert key(a) == key(b) == key(full)
    assert key(mips.search_device(q, top_k=k, nprobe=2)) == key(two)
    assert mips.ivf["nprobe"] == nlist


@pytest.mark.gpu
def test_list_assignment_of_resident_rows_equals_float64_host_assignment():
    """The fused assignment kernel (MFMA GEMM + running arg-max, no score matrix) on fp32 rows and on the int8 rows of a
    resident shard (de-quantised through the shard's LUT while staged): both equal the float64 host assignment, ties to
    the lowest list id (duplicate centroids), nlist not a multiple of the 128-list tile, n not a multiple of 128."""
    import torch
    from densephrases_amd import Shard
    from densephrases_amd.ivf import assign_lists, assign_lists_gpu, assign_lists_resident
    rng = np.random.default_rng(12)
    n, nlist = 20011, 300
    xb, _ = _clustered_db(rng, n, 40)
    cent = O.int8_to_float(xb[rng.choice(n, nlist, replace=False)]).astype(np.float32)
    cent[7] = cent[3]
    want = assign_lists(xb, cent)
    np.testing.assert_array_equal(assign_lists_gpu(xb, cent), want)
    s = Shard(n, device=0)
    s.upload(xb)
    got = assign_lists_resident(s, cent).cpu().numpy()
    np.testing.assert_array_equal(got, want)      # incl. a 4e-8 near-tie between two lists (row 16726): fp64 re-rank
    s.finalize()
    D, I = s.search(O.int8_to_float(xb[:3]), 1)          # the shard is still a working flat shard afterwards
    np.testing.assert_array_equal(I[:, 0], np.arange(3))


@pytest.mark.gpu
def test_kmeans_step_on_the_device_equals_the_host_computation():
    """dph_kmeans_step_dev (what FAISS' IndexIVF::train does for the reference's inner-product indexes,
    build_phrase_index.py:96-142): assignment = arg-max inner product with the current centroids, update = mean of the
    de-quantised members, L2-normalised (spherical); empty lists keep their centroid and report count 0.  Held against a
    float64 host computation; then the whole trainer on a clustered shard: unit-norm centroids, every list used."""
    import torch
    from densephrases_amd import Shard
    from densephrases_amd.ivf import kmeans_sample_size, split_empty_lists, train_centroids_resident
    rng = np.random.default_rng(21)
    n, nlist = 30000, 37
    xb, _ = _clustered_db(rng, n, 24)
    s = Shard(n, device=0)
    s.upload(xb)
    dev = torch.device("cuda", 0)
    x64 = O.int8_to_float(xb).astype(np.float64)
    c0 = O.int8_to_float(xb[rng.choice(n, nlist, replace=False)]).astype(np.float32)
    c0 /= np.linalg.norm(c0, axis=1, keepdims=True)
    c0[5] = c0[2]                                         # a duplicate centroid: list 5 stays empty (ties go to the lower id)
    idx = torch.arange(n, dtype=torch.int64, device=dev)
    sample = torch.empty((n, 768), dtype=torch.int8, device=dev)
    s.gather_rows_dev(idx.data_ptr(), n, sample.data_ptr())
    np.testing.assert_array_equal(sample.cpu().numpy(), xb)
    for spherical in (True, False):
        c = torch.from_numpy(c0.copy()).to(dev)
        assign = torch.empty(n, dtype=torch.int32, device=dev)
        gap = torch.empty(n, dtype=torch.float32, device=dev)
        counts = torch.empty(nlist, dtype=torch.int32, device=dev)
        s.kmeans_step_dev(sample.data_ptr(), n, c.data_ptr(), nlist, assign.data_ptr(), gap.data_ptr(), counts.data_ptr(),
                          spherical=spherical)
        torch.cuda.synchronize()
        sc = x64 @ c0.astype(np.float64).T
        want = np.argmax(sc, 1)
        got = assign.cpu().numpy()
        srt = np.sort(sc, 1)
        clear = (srt[:, -1] - srt[:, -2]) > 1e-3          # fp32 scores: rows with a near-tie may go either way
        np.testing.assert_array_equal(got[clear], want[clear])
        assert (got != want).sum() <= 5
        cnt = counts.cpu().numpy()
        np.testing.assert_array_equal(cnt, np.bincount(got, minlength=nlist))
        assert cnt[5] == 0
        newc = c.cpu().numpy()
        np.testing.assert_array_equal(newc[5], c0[5])
        for l in range(nlist):
            if cnt[l] == 0:
                continue
            m = (xb[got == l].astype(np.int64).sum(0) / cnt[l]) / 20.0 - 2.0          # exact integer sums, then the codec
            if spherical:
                m = m / np.linalg.norm(m)
            np.testing.assert_allclose(newc[l], m, rtol=1e-6, atol=1e-7)
        cc = torch.from_numpy(newc.copy()).to(dev)
        assert split_empty_lists(cc, counts) == 1
        big = int(np.argmax(cnt))
        np.testing.assert_allclose(cc[5].cpu().numpy()[0::2], newc[big][0::2] * (1 + 1 / 1024), rtol=1e-6)
        np.testing.assert_allclose(cc[big].cpu().numpy()[0::2], newc[big][0::2] * (1 - 1 / 1024), rtol=1e-6)
    assert kmeans_sample_size(170_000_000, 4096) == 256 * 4096 and kmeans_sample_size(1000, 8) == 312 and kmeans_sample_size(200, 8) == 200
    with pytest.raises(ValueError):
        kmeans_sample_size(5, 8)
    cent, info = train_centroids_resident(s, 24, iters=8, seed=1, return_info=True)
    assert info["sample_rows"] == 1200 and len(info["seconds_per_iter"]) == 8        # 4 % of 30 000 rows
    np.testing.assert_allclose(np.linalg.norm(cent, axis=1), 1.0, rtol=1e-5)
    from densephrases_amd.ivf import assign_lists
    a = assign_lists(xb, cent)
    assert len(np.unique(a)) >= 20                       # (nearly) every list is in use on the 24-cluster dump


@pytest.mark.gpu
def test_device_side_list_builder_equals_host_builder():
    """dph_index_make_list_major (radix sort by (list, id) + row gather on the GPU) against ivf.build_list_major on the
    host: same stored order (ids of every stored row, padding, tile -> list table through the search results), and the
    IVF search over the device-built shard equals the oracle over the same assignment -- both scans."""
    import torch
    from densephrases_amd import Shard
    from densephrases_amd.ivf import assign_lists, build_list_major, make_list_major_resident, train_centroids
    rng = np.random.default_rng(21)
    n, nlist, nprobe, k = 30011, 40, 5, 10
    xb, centres = _clustered_db(rng, n, 24)
    cent = train_centroids(xb[:8000], nlist, iters=4, seed=2)
    cent[9] = 0.0                                              # an empty list in the middle (score 0 never wins here)
    s = Shard(n, device=0, id_base=700)
    s.upload(xb)
    cent2, assign = make_list_major_resident(s, nlist, centroids=cent)
    assign = assign.cpu().numpy()
    np.testing.assert_array_equal(cent2, cent)
    want = assign_lists(xb, cent)
    np.testing.assert_array_equal(assign, want)
    stored, row_ids, tile_list = build_list_major(xb, assign, nlist, id_base=700)
    s.finalize()
    assert s.ntotal == n
    # every id reconstructs to its own row, wherever the builder put it
    for i in (700, 700 + 17, 700 + n - 1):
        np.testing.assert_array_equal(s.reconstruct(i), O.int8_to_float(xb[i - 700]))
    x = (centres[rng.integers(0, 24, 70)] + rng.normal(0, 0.3, (70, 768))).astype(np.float32)
    for units in (1, 0):
        s.set_tuning("ivf_units", units)
        D, I = s.search_ivf(x, k, nprobe)
        Dr, Ir, D64 = O.ivf_flat_search(x, xb, cent, assign, nprobe, k)
        ok, msg = O.topk_equivalent(D, I, D64, np.where(Ir >= 0, Ir + 700, -1))
        assert ok, msg
        assert s.stats()["uncertified"] == 0
    # the exact search over the permuted shard is still the flat oracle
    Df, If = s.search(x, k)
    Drf, Irf, D64f = O.flat_ip_search(x, xb, k, id_base=700)
    ok, msg = O.topk_equivalent(Df, If, D64f, Irf)
    assert ok, msg
    # and a host-built twin returns the same ids for the same queries
    t = Shard(stored.shape[0], device=0, id_base=700)
    t.upload(stored)
    t.set_row_ids(row_ids, n)
    t.set_ivf(cent, tile_list)
    t.finalize()
    D2, I2 = t.search_ivf(x, k, nprobe)
    np.testing.assert_array_equal(I2, I)
    np.testing.assert_array_equal(D2, D)


@pytest.mark.gpu
def test_list_padding_rows_never_become_candidates():
    """5000 lists of ~8 rows: three quarters of every tile is list padding (all-zero rows, high-digit score 0).  Queries
    with all-negative components score every real row far below 0, so the padding would pass any sampled bound -- 24 dead
    pairs per tile and query row, enough to overflow the scan waves' pair regions and push every row into the retry
    chain.  The scans look such candidates up before emitting them: every row certifies on the first attempt, exact
    search and both IVF scans, and the results are the oracle's."""
    from densephrases_amd.ivf import assign_lists
    rng = np.random.default_rng(55)
    n, nlist, nprobe, n_q, k = 40000, 5000, 100, 64, 10
    xb = O.float_to_int8(rng.normal(0.0, 0.6, (n, 768)).astype(np.float32))
    cent = O.int8_to_float(xb[rng.choice(n, nlist, replace=False)]).astype(np.float32)
    x = -np.abs(rng.normal(0, 0.5, (n_q, 768))).astype(np.float32)
    s, assign = _ivf_shard(xb, cent)
    np.testing.assert_array_equal(assign, assign_lists(xb, cent))
    Df, If = s.search(x, k)
    assert s.stats()["certified_fast"] == n_q, s.stats()
    Drf, Irf, D64f = O.flat_ip_search(x, xb, k)
    ok, msg = O.topk_equivalent(Df, If, D64f, Irf)
    assert ok, msg
    Dr, Ir, D64 = O.ivf_flat_search(x, xb, cent, assign, nprobe, k)
    for units in (0, 1):
        s.set_tuning("ivf_units", units)
        D, I = s.search_ivf(x, k, nprobe)
        assert s.stats()["certified_fast"] == n_q, (units, s.stats())
        ok, msg = O.topk_equivalent(D, I, D64, Ir)
        assert ok, (units, msg)


@pytest.mark.gpu
def test_list_builder_places_every_row_of_a_shard_beyond_2_pow_32_work_items():
    """Regression (round 3): the gather of dph_index_make_list_major ran one wave per row, i.e. n * 64 work-items, and a
    HIP launch silently wraps a global size beyond 2^32 -- on shards of more than 67.1 M rows only the first
    (n * 64 mod 2^32) / 64 sorted rows were gathered, the rest of the permuted shard stayed zero with row_ids = -1
    (every search still "certified": it searched what was there).  70 M rows, lists of very different lengths; rows
    from every part of the sorted order must reconstruct to themselves, the exact search must find planted rows
    wherever they were put, and no stored row of a full tile may be padding."""
    import torch
    from densephrases_amd import Shard
    from densephrases_amd.synth import synthetic_rows
    free = torch.cuda.mem_get_info(0)[0]
    n = 70_000_000 // 32 * 32
    if free < 2.3 * n * 768:
        pytest.skip("needs room for two copies of a 70 M-row shard")
    nlist = 512
    dev = torch.device("cuda", 0)
    s = Shard(n, device=0)
    s.fill_synthetic(seed=42, kind=0)
    g = torch.Generator(device=dev).manual_seed(5)
    u = torch.rand(n, device=dev, generator=g)
    assign = (u * u * nlist).to(torch.int32).clamp_(max=nlist - 1)
    del u
    counts = torch.bincount(assign.to(torch.int64), minlength=nlist).cpu().numpy()
    cent = np.random.default_rng(0).normal(0, 0.5, (nlist, 768)).astype(np.float32)
    s.make_list_major(assign.data_ptr(), cent, stream=torch.cuda.current_stream(dev).cuda_stream)
    s.finalize()
    assert s.ntotal == n and s.n_rows == int(((counts + 31) // 32 * 32).sum())
    a = assign.cpu().numpy()
    del assign
    # ids from the first and the LAST lists (the end of the sorted order is what the wrapped launch never reached)
    rng = np.random.default_rng(1)
    probe = np.concatenate([rng.integers(0, n, 24), np.nonzero(a >= nlist - 3)[0][:12], np.nonzero(a == 0)[0][:4]])
    for i in probe.tolist():
        np.testing.assert_array_equal(s.reconstruct(int(i)), O.int8_to_float(synthetic_rows(int(i), 1, seed=42)[0]))
    rows = np.stack([synthetic_rows(int(i), 1, seed=42)[0] for i in probe[:32]]).astype(np.float32) / 20.0 - 2.0
    x = (rows + rng.normal(0, 0.05, rows.shape)).astype(np.float32)
    D, I = s.search(x, 4)
    np.testing.assert_array_equal(I[:, 0], probe[:32])
    assert s.stats()["uncertified"] == 0


@pytest.mark.gpu
def test_ivf_equals_the_oracle_on_2M_document_ordered_rows_where_recall_is_a_trade_off():
    """Parity, not recall: on 2 M rows of the document-ordered dump (runs of 56..200 near-duplicate rows -- thousands of runs per
    list, so a query's neighbours are NOT all in its best list) with k-means lists built in HBM (256 lists), the kernel's top-10
    under nprobe 1 / 4 / 16 / 256 equals ``oracle.ivf_flat_search`` over exactly the probed lists, id for id, for every query row
    (index.py:52-62: the reference's IVF semantics).  The same run shows the trade-off the SURVEY 8d mixture hides: recall@10
    against the exact search is clearly below 1 at nprobe 1, grows with nprobe, and is exactly 1 when every list is probed."""
    import torch
    from densephrases_amd import Shard
    from densephrases_amd.ivf import make_list_major_resident
    from tests._devdata import gpu_ivf_flat_search
    n, nlist, k, n_q = 2_000_000 // 32 * 32, 256, 10, 96
    s = Shard(n, device=0)
    s.fill_synthetic(seed=21, kind=2)

    class _Rows:
        def __init__(self, ptr):
            self.__cuda_array_interface__ = {"shape": (n, 768), "typestr": "|i1", "data": (int(ptr), False), "version": 2}
    dev = torch.device("cuda", 0)
    xb = torch.as_tensor(_Rows(s.rows_dev_ptr()), device=dev).cpu().numpy().copy()          # the rows in id order, before the builder permutes them
    cent, assign = make_list_major_resident(s, nlist, iters=6, seed=3)
    assign = assign.cpu().numpy()
    s.finalize()
    rng = np.random.default_rng(8)
    pick = rng.integers(0, n, n_q)
    x = (O.int8_to_float(xb[pick]) + rng.normal(0, 0.25, (n_q, 768))).astype(np.float32)
    # exact answer: float64 brute force in plain torch over the rows in id order
    xbt = torch.from_numpy(xb).to(dev)
    q64 = torch.from_numpy(x).to(dev).to(torch.float64)
    best_s = torch.full((n_q, k), -float("inf"), dtype=torch.float64, device=dev)
    best_i = torch.full((n_q, k), -1, dtype=torch.int64, device=dev)
    for r0 in range(0, n, 1 << 18):
        sc = q64 @ (xbt[r0:r0 + (1 << 18)].to(torch.float32) / 20.0 - 2.0).to(torch.float64).T
        ts, ti = torch.topk(sc, k, dim=1)
        cs, ci = torch.cat([best_s, ts], 1), torch.cat([best_i, ti + r0], 1)
        o = torch.topk(cs, k, dim=1)
        best_s, best_i = o.values, torch.gather(ci, 1, o.indices)
    exact = best_i.cpu().numpy()
    recall = {}
    for nprobe in (1, 4, 16, nlist):
        D, I = s.search_ivf(x, k, nprobe)
        assert s.stats()["uncertified"] == 0
        Dr, Ir, D64 = gpu_ivf_flat_search(x, xb, cent, assign, nprobe, k)      # (= O.ivf_flat_search: test_ivf_search_matches_oracle)
        ok, msg = O.topk_equivalent(D, I, D64, Ir)
        assert ok, (nprobe, msg)
        recall[nprobe] = float(np.mean([len(set(a.tolist()) & set(b.tolist())) / k for a, b in zip(I, exact)]))
    assert recall[nlist] == 1.0, recall
    assert recall[1] < 0.99 and recall[1] <= recall[4] <= recall[16] <= 1.0, recall
    s.close()
