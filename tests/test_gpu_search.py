"""GPU parity tests (run with -m gpu on an MI355X): libdph through its C ABI vs the CPU oracle."""
import numpy as np
import pytest

from oracle import mips_oracle as O
from tests._golden import compare_results, load_cases, load_toy_docs

pytestmark = pytest.mark.gpu


def _shard(xb, id_base=0):
    from densephrases_amd import Shard
    s = Shard(xb.shape[0], device=0, id_base=id_base)
    if xb.shape[0]:
        s.upload(xb)
    s.finalize()
    return s


def _rand_db(rng, n):
    """i.i.d. rows ~ float_to_int8(N(0, 0.6^2)).  Large shards come from libdph's own generator on the device (kind 0 =
    40 + 12 z, synth.py; seconds instead of the minutes numpy's float64 normals take), seeded from `rng` so that every test still
    has data of its own; small ones stay numpy draws."""
    if n >= 20000:
        from tests._devdata import device_rows
        return device_rows(n, seed=int(rng.integers(1, 1 << 31)), kind=0)
    return O.float_to_int8(rng.standard_normal((n, 768), dtype=np.float32) * np.float32(0.6))


def _flat(x, xb, k, id_base=0):
    """The oracle of faiss IndexFlatIP.search: numpy (oracle/mips_oracle.py) on shards it finishes in seconds, its torch float64
    restatement on the GPU (tests/_devdata.py, held against the numpy oracle in test_gpu_oracle_restatement_equals_the_numpy_oracle)
    beyond."""
    if xb.shape[0] * x.shape[0] >= 3_000_000:
        from tests._devdata import gpu_flat_ip_search
        return gpu_flat_ip_search(x, xb, k, id_base=id_base)
    return O.flat_ip_search(x, xb, k, id_base=id_base)


def _host_digits(x):
    """numpy replica of dph_quantize_kernel: q ~= sc * (128*q1 + q2)."""
    x = x.astype(np.float32)
    am = np.abs(x).max(1).astype(np.float64)
    s = np.where(am > 0, am / 127.0, 1.0)
    u = x.astype(np.float64) / s[:, None]
    q1 = np.clip(np.rint(u), -127, 127)
    q2 = np.clip(np.rint((u - q1) * 128.0), -64, 64)
    return q1.astype(np.int64), q2.astype(np.int64), s / 128.0


@pytest.mark.parametrize("n_rows,n_q", [(6000, 128), (4099, 7), (33, 3), (5000, 200), (700, 256)])
def test_scan_buckets_hold_exact_integer_scores_cold(n_rows, n_q):
    """Filter scan without a bound + refine: every row of the shard is in every query row's bucket exactly once, with
    the exact integer score 128*<q1,n> + <q2,n> (n_q > 128 runs the 256-row kernel)."""
    rng = np.random.default_rng(n_rows)
    xb = _rand_db(rng, n_rows)
    x = rng.normal(0, 0.5, (n_q, 768)).astype(np.float32)
    s = _shard(xb)
    buckets, lost = s.debug_scan_buckets(x)
    assert not lost.any()
    q1, q2, _ = _host_digits(x)
    ref = 128 * (q1 @ xb.astype(np.int64).T) + q2 @ xb.astype(np.int64).T            # [n_q, N] exact
    for q in range(n_q):
        score, rows = buckets[q]
        assert rows.size == n_rows, f"q{q}: {rows.size} keys for {n_rows} rows"
        assert np.array_equal(np.sort(rows), np.arange(n_rows)), f"q{q}: rows missing / duplicated / out of range"
        np.testing.assert_array_equal(score.astype(np.int64), ref[q, rows.astype(np.int64)], err_msg=f"q{q}")


@pytest.mark.parametrize("n_rows,n_q,stride", [(70000, 128, 1), (70000, 256, 1), (70000, 130, 7)])
def test_scan_buckets_under_a_bound(n_rows, n_q, stride):
    """Filter scan under per-row bounds tau: a visited row is emitted iff 128*H + lmax > tau (so every row with
    I > tau is there), every emitted key carries the exact integer score, nothing appears twice."""
    rng = np.random.default_rng(n_rows + n_q)
    xb = _rand_db(rng, n_rows)
    x = rng.normal(0, 0.5, (n_q, 768)).astype(np.float32)
    s = _shard(xb)
    q1, q2, _ = _host_digits(x)
    xi = xb.astype(np.int64)
    H = q1 @ xi.T
    ref = 128 * H + q2 @ xi.T
    visited = np.zeros(n_rows, bool)
    for t in range(0, (n_rows + 31) // 32, stride):
        visited[t * 32:(t + 1) * 32] = True
    tau = np.sort(ref[:, visited], axis=1)[:, -40].astype(np.int32)        # ~40 rows beat it per query row
    tau[0] = np.iinfo(np.int32).min                                       # no bound for row 0: everything visited
    buckets, lost = s.debug_scan_buckets(x, tau=tau, tile_stride=stride)
    lmax = s.debug_lmax(n_q).astype(np.int64)
    for q in range(n_q):
        score, rows = buckets[q]
        r = rows.astype(np.int64)
        assert len(set(r.tolist())) == r.size and visited[r].all()
        np.testing.assert_array_equal(score.astype(np.int64), ref[q, r])
        if q == 0:
            assert r.size == min(int(visited.sum()), 32768)       # cold row: everything visited, up to the bucket's capacity
            continue
        assert not lost[q]
        must = np.nonzero(visited & (128 * H[q] + lmax[q] > int(tau[q])))[0]
        assert np.array_equal(np.sort(r), must), f"q{q}: emitted set differs from the filter's definition"
        assert (ref[q, must] > tau[q]).sum() >= 39


@pytest.mark.parametrize("n_rows,n_q,k", [(50000, 128, 10), (50000, 2, 10), (9001, 130, 10), (300, 5, 20),
                                          (31, 4, 10), (5, 3, 10), (1, 2, 3), (20000, 64, 100), (30000, 6, 200),
                                          (60000, 256, 10), (60000, 300, 10), (40000, 512, 10), (120000, 9, 1024)])
def test_search_matches_oracle(n_rows, n_q, k):
    rng = np.random.default_rng(n_rows * 7 + n_q)
    xb = _rand_db(rng, n_rows)
    x = rng.normal(0, 0.5, (n_q, 768)).astype(np.float32)
    planted = rng.integers(0, n_rows, n_q // 2 + 1)
    x[:planted.size] = (xb[planted].astype(np.float32) / 20 - 2 + rng.normal(0, 0.1, (planted.size, 768))).astype(np.float32)
    s = _shard(xb, id_base=1000)
    D, I = s.search(x, k)
    Dr, Ir, D64 = _flat(x, xb, k, id_base=1000)
    ok, msg = O.topk_equivalent(D, I, D64, Ir)
    assert ok, msg
    np.testing.assert_array_equal(I[:planted.size, 0], planted + 1000)
    st = s.stats()
    assert st["rows"] == n_q and st["uncertified"] == 0


def test_gpu_oracle_restatement_equals_the_numpy_oracle():
    """tests/_devdata.py's torch float64 restatements of the oracle (used where numpy takes minutes) against the numpy oracle itself:
    ids identical -- exact ties (duplicate rows) in id order, fewer rows than k padded alike -- scores to float64 rounding."""
    from tests._devdata import gpu_flat_ip_search
    rng = np.random.default_rng(12)
    xb = _rand_db(rng, 300_000)
    xb[1000:1040] = xb[7]                                # 41 exact ties
    x = rng.normal(0, 0.5, (9, 768)).astype(np.float32)
    x[0] = xb[7].astype(np.float32) / 20 - 2
    for k, sub in ((10, xb), (100, xb[:50_000]), (20, xb[:13])):
        Dn, In, D64n = O.flat_ip_search(x, sub, k, id_base=300)
        Dg, Ig, D64g = gpu_flat_ip_search(x, sub, k, id_base=300)
        np.testing.assert_array_equal(Ig, In)
        np.testing.assert_array_equal(Dg, Dn)
        np.testing.assert_allclose(D64g[In >= 0], D64n[In >= 0], rtol=1e-13, atol=1e-9)


def test_search_empty_shard_and_zero_queries():
    s = _shard(np.zeros((0, 768), np.int8))
    D, I = s.search(np.ones((2, 768), np.float32), 5)
    assert (I == -1).all() and (D == -O.FLT_MAX).all()
    s2 = _shard(_rand_db(np.random.default_rng(0), 64))
    D, I = s2.search(np.zeros((0, 768), np.float32), 5)
    assert D.shape == (0, 5)
    D, I = s2.search(np.zeros((1, 768), np.float32), 3)          # all-zero query: every score 0, ids 0,1,2
    np.testing.assert_array_equal(I[0], [0, 1, 2])


@pytest.mark.parametrize("n_dup", [40, 3000])
def test_duplicate_rows_and_the_retry_chain(n_dup):
    """Exact copies of the best row.  40 copies fit the candidates the first attempt re-scores (certified at once, ids
    in ascending order); 3000 copies exceed even the retry's 2048 candidates, so the row must go through the fp64 full
    scan -- and still return the ten lowest-id copies."""
    rng = np.random.default_rng(5)
    n_rows = 400000
    xb = _rand_db(rng, n_rows)
    x = rng.normal(0, 0.5, (4, 768)).astype(np.float32)
    hot = xb[123].copy()
    x[0] = hot.astype(np.float32) / 20 - 2
    dup = rng.choice(np.arange(200, n_rows), n_dup, replace=False)
    xb[dup] = hot
    s = _shard(xb)
    D, I = s.search(x, 10)
    Dr, Ir, D64 = _flat(x, xb, 10)
    ok, msg = O.topk_equivalent(D, I, D64, Ir)
    assert ok, msg
    np.testing.assert_array_equal(I[0], Ir[0])     # exact ties: lowest ids first
    st = s.stats()
    assert st["uncertified"] == 0
    if n_dup == 40:
        assert st["certified_fast"] == 4
    else:
        assert st["certified_fast"] == 3 and st["exact_fallback"] == 1


def test_lost_pairs_are_repaired_by_the_on_device_retry():
    """No pre-pass at all (ladder {0}) on a shard far too big for a cold scan: every bucket overflows, the first attempt
    certifies nothing, and the retry -- a scan under the bound derived from the k-th best integer score of what WAS
    seen -- returns the exact answer without the fp64 fallback."""
    rng = np.random.default_rng(15)
    n_rows, n_q = 400000, 20
    xb = _rand_db(rng, n_rows)
    x = rng.normal(0, 0.5, (n_q, 768)).astype(np.float32)
    s = _shard(xb)
    s.set_tuning("ladder", 0)
    D, I = s.search(x, 10)
    Dr, Ir, D64 = _flat(x, xb, 10)
    ok, msg = O.topk_equivalent(D, I, D64, Ir)
    assert ok, msg
    st = s.stats()
    assert st["certified_fast"] == 0 and st["certified_wide"] == n_q and st["uncertified"] == 0


def test_mixture_dump_with_saturated_outlier_rows():
    """The kind-1 synthetic dump (mixture of 4096 Gaussians + saturated rows): the outlier rows are taken out of the
    certificate's row-norm bound (rmax well below the true maximum) yet still found when they are the answer, and the
    search agrees with the CPU oracle over the host replica of the dump."""
    from densephrases_amd import Shard
    from densephrases_amd.synth import synthetic_outlier_rows, synthetic_rows
    n_rows, seed = 1_200_000, 42
    out_rows = synthetic_outlier_rows(n_rows, seed)
    assert out_rows.size >= 1
    s = Shard(n_rows, device=0)
    s.fill_synthetic(seed=seed, kind=1)
    s.finalize()
    ss = s.shard_stats()
    # the cut sits at a histogram bin edge: the saturated rows plus at most a few hundred of the largest ordinary rows
    assert out_rows.size <= ss["n_outliers"] <= 1024 and ss["rmax"] < 0.7 * ss["rmax_all"]
    from tests._devdata import device_rows
    xb = device_rows(n_rows, seed=seed, kind=1)          # (= synth.synthetic_rows: test_synthetic_fill_matches_host_generator)
    np.testing.assert_array_equal(xb[out_rows[0]], synthetic_rows(int(out_rows[0]), 1, seed, kind=1)[0])
    rng = np.random.default_rng(3)
    x = rng.normal(0, 0.5, (24, 768)).astype(np.float32)
    x[0] = xb[out_rows[0]].astype(np.float32) / 20 - 2                      # the answer IS an outlier row
    x[1] = -x[0]
    planted = rng.integers(0, n_rows, 8)
    x[2:10] = xb[planted].astype(np.float32) / 20 - 2 + rng.normal(0, 0.1, (8, 768)).astype(np.float32)
    x[10:16] = xb[planted[:6]].astype(np.float32) / 20 - 2 + rng.normal(0, 0.6, (6, 768)).astype(np.float32)   # inside a cluster
    D, I = s.search(x, 10)
    assert I[0, 0] == out_rows[0]
    Dr, Ir, D64 = _flat(x, xb, 10)
    ok, msg = O.topk_equivalent(D, I, D64, Ir)
    assert ok, msg
    st = s.stats()
    assert st["uncertified"] == 0 and st["certified_fast"] >= 20


def test_reconstruct_and_id2docword():
    rng = np.random.default_rng(2)
    xb = _rand_db(rng, 100)
    s = _shard(xb, id_base=50)
    np.testing.assert_array_equal(s.reconstruct(57), O.int8_to_float(xb[7]))
    from densephrases_amd import DphError
    with pytest.raises(DphError):
        s.reconstruct(49)
    s.set_idx2id(np.arange(100, dtype=np.int32) // 10, np.arange(100, dtype=np.int32) % 10)
    doc, word = s.id2docword(np.array([[50, 149], [-1, 9999]]))
    np.testing.assert_array_equal(doc, [[0, 9], [0, 9]])
    np.testing.assert_array_equal(word, [[0, 9], [0, 9]])


@pytest.mark.parametrize("kind", [0, 1, 2, 3, 4])
def test_synthetic_fill_matches_host_generator(kind):
    """every on-device generator (i.i.d., mixture + outliers, document-ordered runs, mixture) = its host replica, byte for
    byte (the id base shifts the global row index, runs and clusters follow it)"""
    from densephrases_amd import Shard
    from densephrases_amd.synth import synthetic_rows
    s = Shard(5000, device=0, id_base=777)
    s.fill_synthetic(seed=42, kind=kind)
    s.finalize()
    want = synthetic_rows(777, 5000, seed=42, kind=kind)
    for r in (0, 1, 31, 32, 255, 256, 300, 4999):
        np.testing.assert_array_equal(s.reconstruct(777 + r), O.int8_to_float(want[r]))


def test_window_rescore_matches_oracle():
    docs = load_toy_docs()
    index = O.build_index_from_docs(docs)
    from densephrases_amd import DocMeta, DocStore, MIPS
    store = DocStore([DocMeta(m.doc_idx, m.title, m.context, m.f2o_start, m.word2char_start, m.word2char_end, m.start)
                      for m in docs])
    mips = MIPS.from_store(store)
    rng = np.random.default_rng(9)
    B, k = 5, 7
    ids = rng.integers(0, index.ntotal, (B, k))
    ids[0, 0], ids[0, 1] = 0, index.ntotal - 1                  # windows that run off both ends of the shard
    doc, word = O.get_idxs(index, ids)
    first = rng.normal(50, 5, (B, k)).astype(np.float32)
    q = rng.normal(0, 0.5, (B, 768)).astype(np.float32)
    for direction, name in ((0, "end"), (1, "start")):
        for L in (1, 3, 10):
            pred, best, arg, vecs = mips.shard.rescore(direction, q, k, L, ids, doc, word, first, want_vecs=True)
            p2, b2, sc2, v2, a2 = O.window_rescore(index, np.repeat(q, k, 0), doc.reshape(-1), word.reshape(-1),
                                                   ids.reshape(-1), first.reshape(-1), L, name, branch="ram")
            np.testing.assert_array_equal(pred, p2)
            np.testing.assert_array_equal(arg, a2)
            np.testing.assert_allclose(best, b2, rtol=1e-6, atol=1e-4)
            own = v2[:, 0] if direction == 0 else v2[:, -1]
            np.testing.assert_array_equal(vecs[:, 0], own)
            np.testing.assert_array_equal(vecs[:, 1], v2[np.arange(B * k), a2])


CASES, VECS = load_cases()


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_mips_matches_reference_golden(ci):
    """The product MIPS class against outputs of the reference's own index.py (tests/golden)."""
    c = CASES[ci]
    if c["return_idxs"] and c["branch"] == "hdf5":
        pytest.skip("reference HDF5 branch returns raw int8 for the candidate's own vector (index.py:263-272); "
                    "the resident-shard path follows the RAM branch")
    from densephrases_amd import DocMeta, DocStore, MIPS
    docs = load_toy_docs()
    store = DocStore([DocMeta(m.doc_idx, m.title, m.context, m.f2o_start, m.word2char_start, m.word2char_end, m.start)
                      for m in docs])
    mips = MIPS.from_store(store)
    got = mips.search(c["query_arr"].astype(np.float64), q_texts=[f"q{i}" for i in range(c["B"])], top_k=c["top_k"],
                      aggregate=c["aggregate"], return_idxs=c["return_idxs"], max_answer_length=c["L"],
                      agg_strat=c["agg_strat"], return_sent=c["return_sent"])
    compare_results(got, c["results"], VECS, case=c)
    dense = mips.search_dense(c["query_arr"], top_k=c["top_k"])
    for a, b in zip(dense, c["dense"]):
        b = np.asarray(b)
        if b.dtype.kind == "f":
            np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-5)
        else:
            np.testing.assert_array_equal(a, b)


def test_outlier_rows_do_not_set_the_sampled_bound():
    """A shard whose best rows for every query are OUTLIERS (1 % saturated rows: every component at the int8 limits, norm 3547 against
    ~ 1100; set aside at finalize and scored against every query in full, dph_outlier_kernel).  They sit in every ladder level's bucket
    in full, not as a 1-in-stride sample, so they must not count towards the kp-th best sampled score (dph_threshold_kernel): with
    them the bound was the kp-th outlier's score -- kp = 16 rows above it in the whole shard -- and for k > kp no first attempt could
    certify (k = 500: 25 of 129 rows settled before the fp64 scan; at 170 M rows the retry drowned and dph_search returned
    DPH_E_UNCERTIFIED).  Now every row is certified by the first attempt, for k on both sides of the outlier count, and exact."""
    from densephrases_amd import Shard
    from tests._devdata import gpu_flat_ip_search
    rng = np.random.default_rng(21)
    n = 70001
    xb = _rand_db(rng, n)
    sat = rng.choice(n, 700, replace=False)
    xb[sat] = rng.choice(np.array([-128, 127], np.int8), (len(sat), 768))
    s = Shard(n, device=0)
    s.upload(xb)
    s.finalize()
    assert 500 <= s.shard_stats()["n_outliers"] <= 1024
    x = rng.normal(0, 0.5, (129, 768)).astype(np.float32)
    for k in (10, 100, 500, 1024):
        D, I = s.search(x, k)
        st = s.stats()
        assert st["certified_fast"] == len(x) and st["uncertified"] == 0, (k, st)
        Dr, Ir, D64 = gpu_flat_ip_search(x, xb, k)
        ok, msg = O.topk_equivalent(D, I, D64, Ir)
        assert ok, (k, msg)
        assert np.isin(I[:, :10], sat).all()                          # the best rows ARE the saturated ones
    s.close()


def test_a_dropped_mips_gives_its_hbm_back():
    """A MIPS that is no longer referenced is collected, and its shard's device memory with it (the fuzz soak ran 904 MIPS objects
    into hipErrorOutOfMemory: the C++ host half held a callback that held the MIPS -- a cycle through an extension type, which
    the collector cannot see); MIPS.close() releases it at once."""
    import gc
    import weakref
    import torch
    from densephrases_amd import DocMeta, DocStore, MIPS
    docs = load_toy_docs()
    q = CASES[0]["query_arr"].astype(np.float64)

    def make():
        m = MIPS.from_store(DocStore([DocMeta(d.doc_idx, d.title, d.context, d.f2o_start, d.word2char_start, d.word2char_end, d.start)
                                      for d in docs]))
        m.search(q, top_k=3)                              # (builds the host half and the per-shape search state)
        return m

    make().close()                                        # warm-up: allocator pools, code objects
    gc.collect()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    m = make()
    assert torch.cuda.mem_get_info(0)[0] < free0 - (64 << 20)            # a shard's scratch is hundreds of MiB
    ref = weakref.ref(m)
    del m
    gc.collect()
    assert ref() is None, "the MIPS object is still alive: something holds it"
    assert torch.cuda.mem_get_info(0)[0] >= free0 - (16 << 20)
    m = make()
    m.close()
    assert torch.cuda.mem_get_info(0)[0] >= free0 - (16 << 20)
    m.close()                                             # idempotent


def test_large_synthetic_shard_properties():
    """N = 1M rows generated on the device: planted queries come back first, the answer is invariant under the
    certificate path taken, and a 16-row slice agrees with the full CPU oracle."""
    from densephrases_amd import Shard
    from densephrases_amd.synth import synthetic_rows
    n_rows = 1_000_000
    s = Shard(n_rows, device=0)
    s.fill_synthetic(seed=42)
    s.finalize()
    rng = np.random.default_rng(11)
    planted = rng.integers(0, n_rows, 128)
    rows = np.stack([synthetic_rows(int(p), 1, 42)[0] for p in planted])
    x = (rows.astype(np.float32) / 20 - 2 + rng.normal(0, 0.1, rows.shape)).astype(np.float32)
    D, I = s.search(x, 10)
    np.testing.assert_array_equal(I[:, 0], planted)
    assert (np.diff(D, axis=1) <= 0).all()
    assert s.stats()["certified_fast"] == 128
    xb = np.empty((n_rows, 768), np.int8)
    for r0 in range(0, n_rows, 100_000):
        xb[r0:r0 + 100_000] = synthetic_rows(r0, 100_000, 42)
    Dr, Ir, D64 = _flat(x[:16], xb, 10)
    ok, msg = O.topk_equivalent(D[:16], I[:16], D64, Ir)
    assert ok, msg


def test_two_shards_on_one_device_match_single_shard():
    """The multi-GPU data path minus the collective: two range shards (doc-aligned cut, global ids via id_base) on
    one device, their records packed exactly like the all-gather buffer, merged by dph_merge_topk_dev, and the
    window results followed back through `src` -- must equal the single-shard search + window re-score."""
    import torch
    from densephrases_amd import Shard
    from densephrases_amd.dist import RecordLayout, ShardedSearcher, exchange_and_merge, partition_rows
    rng = np.random.default_rng(21)
    n_rows, B, k, L, doc_len = 40000, 8, 10, 10, 50
    xb = _rand_db(rng, n_rows)
    xb[n_rows // 2 + 7] = xb[11]                                    # a cross-shard exact tie
    doc = (np.arange(n_rows) // doc_len).astype(np.int32)
    word = (np.arange(n_rows) % doc_len).astype(np.int32)
    doc_ids = np.arange(n_rows // doc_len, dtype=np.int32)
    f2o_off = np.arange(0, n_rows + 1, doc_len, dtype=np.int64)
    f2o = np.tile(np.arange(doc_len, dtype=np.int32) * 2, n_rows // doc_len)     # gaps of 2: masks are exercised
    q = rng.normal(0, 0.5, (B, 1536)).astype(np.float32)
    q[0, :768] = xb[11].astype(np.float32) / 20 - 2
    dev = torch.device("cuda", 0)

    def make(lo, hi):
        s = Shard(hi - lo, device=0, id_base=lo)
        s.upload(xb[lo:hi])
        s.set_idx2id(doc[lo:hi], word[lo:hi])
        s.set_f2o(doc_ids, f2o_off, f2o)
        s.finalize()
        return s

    full = ShardedSearcher(make(0, n_rows), B, k, L, device=dev)
    want = {kk: v.clone() for kk, v in full.step(torch.from_numpy(q).to(dev)).items()}
    parts = partition_rows(n_rows, 2, align=doc_len)
    assert parts[0][1] % doc_len == 0
    searchers = [ShardedSearcher(make(lo, hi), B, k, L, device=dev) for lo, hi in parts]
    layout = RecordLayout(2 * B, k)
    rec_all = torch.zeros((2, layout.nbytes), dtype=torch.uint8, device=dev)
    for r, s in enumerate(searchers):
        s.step(torch.from_numpy(q).to(dev))
        rec_all[r].copy_(s.rec)
    m = searchers[0]
    m.world = 2                                                      # merge kernel reads two parts

    class _NoDist:                                                   # the gather already happened above
        @staticmethod
        def all_gather_into_tensor(out, inp):
            pass

    D, I, best, pred, status = exchange_and_merge(layout, m.rec, rec_all, _NoDist, 2, m._merge)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(I.cpu().numpy(), want["I"].cpu().numpy())
    np.testing.assert_array_equal(D.cpu().numpy(), want["D"].cpu().numpy())
    np.testing.assert_array_equal(pred.cpu().numpy(), want["pred"].cpu().numpy())
    np.testing.assert_allclose(best.cpu().numpy(), want["best"].cpu().numpy(), rtol=0, atol=0)
    assert int(status.max()) == 0
    assert I.cpu().numpy()[0, 0] == 11 and I.cpu().numpy()[0, 1] == n_rows // 2 + 7    # tie: lower id first

    # the unfused form (dph_merge_topk_dev + following `src` in torch) gives the same five arrays
    from densephrases_amd import _lib
    Dg, Ig = torch.empty_like(D), torch.empty_like(I)
    src = torch.empty((2 * B, k), dtype=torch.int32, device=dev)

    def unfused(va):
        _lib.merge_topk_dev(0, va["D"].data_ptr(), va["I"].data_ptr(), 2, 2 * B, k, Dg.data_ptr(), Ig.data_ptr(),
                            src.data_ptr(), part_stride_bytes=layout.nbytes)
        return Dg, Ig, src

    D2, I2, best2, pred2, status2 = exchange_and_merge(layout, m.rec, rec_all, _NoDist, 2, unfused)
    torch.cuda.synchronize()
    for a, b in ((D, D2), (I, I2), (best, best2), (pred, pred2), (status, status2)):
        np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())


def test_mips_from_reference_layout_files(tmp_path):
    """The reference's constructor arguments end to end: phrase/*.hdf5 + idx2id.hdf5 written by h5py (python3.9 of this
    image) in the reference's layout, read back through libhdf5/ctypes, searched on the GPU, compared with the golden
    output of the reference's own index.py."""
    import os
    import subprocess
    py39 = "/opt/conda/bin/python3.9"
    if not os.path.exists(py39):
        pytest.skip("no interpreter with h5py to write the fixture")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([py39, os.path.join(here, "_make_h5_dump.py"), os.path.join(here, "golden", "toy_dump.npz"),
                        str(tmp_path)], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("h5py writer failed: " + r.stderr[-200:])
    from densephrases_amd import MIPS
    idx_dir = os.path.join(str(tmp_path), "start", "toy_flat_none")
    mips = MIPS(phrase_dump_dir=os.path.join(str(tmp_path), "phrase"), index_path=os.path.join(idx_dir, "index.faiss"),
                idx2id_path=os.path.join(idx_dir, "idx2id.hdf5"), cuda=True)
    assert mips.index.ntotal == 261 and mips.index.d == 768
    c = CASES[1]                                  # hdf5 branch, aggregate opt1
    got = mips.search(c["query_arr"].astype(np.float64), q_texts=[f"q{i}" for i in range(c["B"])], top_k=c["top_k"],
                      aggregate=c["aggregate"], max_answer_length=c["L"], agg_strat=c["agg_strat"])
    compare_results(got, c["results"], VECS)
    # the packed copy of the row range (cache_dir): written by one start, streamed by the next -- same shard, same answer
    cache = os.path.join(str(tmp_path), "packed")
    for turn in range(2):
        again = MIPS(phrase_dump_dir=os.path.join(str(tmp_path), "phrase"), index_path=os.path.join(idx_dir, "index.faiss"),
                     idx2id_path=os.path.join(idx_dir, "idx2id.hdf5"), cuda=True, cache_dir=cache)
        assert os.path.exists(os.path.join(cache, "rows_0_261.json"))
        if turn == 1:
            assert again.store._cache is None and not [f for f in os.listdir(cache) if "tmp" in f]
        got = again.search(c["query_arr"].astype(np.float64), q_texts=[f"q{i}" for i in range(c["B"])], top_k=c["top_k"],
                           aggregate=c["aggregate"], max_answer_length=c["L"], agg_strat=c["agg_strat"])
        compare_results(got, c["results"], VECS)


def test_device_step_settles_every_row_without_the_host():
    """The device-resident loop (ShardedSearcher.step -> dph_search_dev) on a shard whose best row has 3000 exact
    copies: the first attempt and the retry cannot certify it, the on-device fp64 scan does -- status is 0 for every
    row when step() returns, and the answer (incl. the window results) equals the oracle's."""
    import torch
    from densephrases_amd import Shard
    from densephrases_amd.dist import ShardedSearcher
    rng = np.random.default_rng(8)
    n_rows, B, k, L = 400000, 4, 10, 5
    xb = _rand_db(rng, n_rows)
    hot = xb[777].copy()
    dup = rng.choice(np.arange(1000, n_rows), 3000, replace=False)
    xb[dup] = hot
    q = rng.normal(0, 0.5, (B, 1536)).astype(np.float32)
    q[0, :768] = hot.astype(np.float32) / 20 - 2
    s = Shard(n_rows, device=0)
    s.upload(xb)
    s.set_idx2id((np.arange(n_rows) // 100).astype(np.int32), (np.arange(n_rows) % 100).astype(np.int32))
    s.set_f2o(np.arange(n_rows // 100, dtype=np.int32), np.arange(0, n_rows + 1, 100, dtype=np.int64),
              np.tile(np.arange(100, dtype=np.int32), n_rows // 100))
    s.finalize()
    ss = ShardedSearcher(s, B, k, L, device=torch.device("cuda", 0))
    out = ss.step(torch.from_numpy(q).cuda())
    assert int((out["status"] != 0).sum()) == 0
    st = s.stats()
    assert st["exact_fallback"] == 1 and st["uncertified"] == 0
    stacked = np.concatenate([q[:, :768], q[:, 768:]], 0)
    Dr, Ir, D64 = _flat(stacked, xb, k)
    ok, msg = O.topk_equivalent(out["D"].cpu().numpy(), out["I"].cpu().numpy(), D64, Ir)
    assert ok, msg
    np.testing.assert_array_equal(out["I"].cpu().numpy()[0], Ir[0])          # the ten lowest-id copies, in id order


@pytest.mark.parametrize("ci", [i for i, c in enumerate(CASES) if not c["return_idxs"]])
def test_mips_device_and_stream_forms_match_reference_golden(ci):
    """search_device (query tensor already on the GPU) and search_stream (GPU half of batch t+1 overlapped with the
    host half of batch t) return what the reference returned for the same queries."""
    import torch
    c = CASES[ci]
    from densephrases_amd import DocMeta, DocStore, MIPS
    docs = load_toy_docs()
    store = DocStore([DocMeta(m.doc_idx, m.title, m.context, m.f2o_start, m.word2char_start, m.word2char_end, m.start)
                      for m in docs])
    mips = MIPS.from_store(store)
    texts = [f"q{i}" for i in range(c["B"])]
    kw = dict(top_k=c["top_k"], aggregate=c["aggregate"], max_answer_length=c["L"], agg_strat=c["agg_strat"],
              return_sent=c["return_sent"])
    q_dev = torch.from_numpy(c["query_arr"].astype(np.float32)).cuda()
    compare_results(mips.search_device(q_dev, q_texts=texts, **kw), c["results"], VECS, case=c)
    outs = list(mips.search_stream([c["query_arr"], q_dev, c["query_arr"]], q_texts=[texts] * 3, **kw))
    assert len(outs) == 3
    for got in outs:
        compare_results(got, c["results"], VECS, case=c)


@pytest.mark.parametrize("fine_stride", [None, 4])
def test_clustered_rows_and_saturated_codes(fine_stride):
    """Non-i.i.d. data: 48 tight clusters (many near-equal top scores for a query at a cluster centre), rows clipped to
    the int8 range ends (-128 / 127 codes present), one all-zero query and one huge-norm query.  Exercises the
    threshold pre-pass (a finer last level in the second variant), the outlier-row cut and the certificate on a score
    distribution unlike the synthetic benchmark's."""
    rng = np.random.default_rng(99)
    n_rows, n_q, k = 300000, 40, 10
    centres = rng.normal(0, 0.9, (48, 768)).astype(np.float32)
    a = rng.integers(0, 48, n_rows)
    xf = centres[a] + rng.normal(0, 0.12, (n_rows, 768)).astype(np.float32)
    xf[::1000] *= 6.0                                              # rows that saturate the codec at both ends
    xb = O.float_to_int8(xf)
    assert xb.min() == -128 and xb.max() == 127
    x = (centres[rng.integers(0, 48, n_q)] + rng.normal(0, 0.02, (n_q, 768))).astype(np.float32)
    x[0] = 0.0
    x[1] *= 1e4
    x[2] = -x[2]
    s = _shard(xb)
    if fine_stride is not None:
        s.set_tuning("fine_stride", fine_stride)
    D, I = s.search(x, k)
    Dr, Ir, D64 = _flat(x, xb, k)
    ok, msg = O.topk_equivalent(D, I, D64, Ir)
    assert ok, msg
    assert s.stats()["uncertified"] == 0


def test_merge_records_padding_ties_and_status():
    """dph_merge_records_dev on dense per-field arrays: FAISS-style padding (-1 ids) in the parts and in the output,
    exact score ties across parts (lower id first), winners carry their part's window results, status = max."""
    import torch
    from densephrases_amd import _lib
    rng = np.random.default_rng(5)
    P, n, k = 3, 6, 4
    D = np.full((P, n, k), -3.4028234663852886e38, np.float32)
    I = np.full((P, n, k), -1, np.int64)
    best = rng.normal(0, 1, (P, n, k))
    pred = rng.integers(0, 100, (P, n, k)).astype(np.int32)
    status = rng.integers(0, 2, (P, n)).astype(np.int32)
    for p in range(P):
        for r in range(n):
            m = int(rng.integers(0, k + 1)) if r else 0            # row 0: no valid entry anywhere -> all padding
            sc = np.sort(rng.integers(0, 6, m).astype(np.float32))[::-1]   # few distinct scores: ties across parts
            D[p, r, :m] = sc
            I[p, r, :m] = rng.choice(1000, m, replace=False) + 1000 * p
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(a).to(dev)                      # noqa: E731
    Dd, Id, bd, pd, sd = t(D), t(I), t(best), t(pred), t(status)
    Do = torch.empty((n, k), dtype=torch.float32, device=dev)
    Io = torch.empty((n, k), dtype=torch.int64, device=dev)
    bo = torch.empty((n, k), dtype=torch.float64, device=dev)
    po = torch.empty((n, k), dtype=torch.int32, device=dev)
    so = torch.empty((n,), dtype=torch.int32, device=dev)
    _lib.merge_records_dev(0, Dd.data_ptr(), Id.data_ptr(), bd.data_ptr(), pd.data_ptr(), sd.data_ptr(), P, n, k,
                           Do.data_ptr(), Io.data_ptr(), bo.data_ptr(), po.data_ptr(), so.data_ptr())
    torch.cuda.synchronize()
    for r in range(n):
        cand = [(-float(D[p, r, c]), int(I[p, r, c]), p, c) for p in range(P) for c in range(k) if I[p, r, c] >= 0]
        cand.sort()
        for j in range(k):
            if j < len(cand):
                _, i, p, c = cand[j]
                assert int(Io[r, j]) == i and float(Do[r, j]) == float(D[p, r, c])
                assert float(bo[r, j]) == best[p, r, c] and int(po[r, j]) == pred[p, r, c]
            else:
                assert int(Io[r, j]) == -1 and float(bo[r, j]) == -1e9 and int(po[r, j]) == -1
        assert int(so[r]) == status[:, r].max()


@pytest.mark.parametrize("levels", [None, (16, 2)])
def test_union_bound_two_phase_search_matches_single_shard(levels):
    """The two-phase sharded search (dph_search_sample_dev -> union of the shards' samples -> dph_search_bounded_dev ->
    certificate after the merge) on two shards of one device equals the single-shard search + window re-score, also
    when a shard holds NO row above the union bound for a query (it returns padding and a bound; status 2)."""
    import torch
    from densephrases_amd import Shard
    from densephrases_amd.dist import RecordLayout, ShardedSearcher, exchange_and_merge, partition_rows
    rng = np.random.default_rng(77)
    n_rows, B, k, L, doc_len = 600000, 8, 10, 10, 100
    xb = _rand_db(rng, n_rows)
    hot = O.float_to_int8(rng.normal(0.0, 1.5, (1, 768)).astype(np.float32))[0]
    where = rng.choice(np.arange(100, n_rows // 2), 2000, replace=False)   # a cluster that lives in the first shard only
    xb[where] = np.clip(hot[None, :].astype(np.int32) + rng.integers(-2, 3, (2000, 768)), -128, 127).astype(np.int8)
    xb[n_rows // 2 + 7] = xb[11]                                      # a cross-shard exact tie
    doc = (np.arange(n_rows) // doc_len).astype(np.int32)
    word = (np.arange(n_rows) % doc_len).astype(np.int32)
    doc_ids = np.arange(n_rows // doc_len, dtype=np.int32)
    f2o_off = np.arange(0, n_rows + 1, doc_len, dtype=np.int64)
    f2o = np.tile(np.arange(doc_len, dtype=np.int32), n_rows // doc_len)
    q = rng.normal(0, 0.5, (B, 1536)).astype(np.float32)
    q[0, :768] = xb[11].astype(np.float32) / 20 - 2                   # answer: the tie, lower id first
    q[1, :768] = xb[n_rows - 5].astype(np.float32) / 20 - 2           # answer in the second shard
    q[2, :768] = hot.astype(np.float32) / 20 - 2                      # every good row is in the first shard
    q[2, 768:] = hot.astype(np.float32) / 20 - 2
    dev = torch.device("cuda", 0)

    def make(lo, hi):
        s = Shard(hi - lo, device=0, id_base=lo)
        s.upload(xb[lo:hi])
        s.set_idx2id(doc[lo:hi], word[lo:hi])
        s.set_f2o(doc_ids, f2o_off, f2o)
        s.finalize()
        if levels is not None:
            s.set_tuning("ladder", *levels)
        return s

    qd = torch.from_numpy(q).to(dev)
    full = ShardedSearcher(make(0, n_rows), B, k, L, device=dev)
    want = {kk: v.clone() for kk, v in full.step(qd).items()}
    assert int(want["status"].max()) == 0
    parts = partition_rows(n_rows, 2, align=doc_len)
    ss = [ShardedSearcher(make(lo, hi), B, k, L, device=dev, union_bounds=True) for lo, hi in parts]
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
    tau = [s.tau.cpu().numpy() for s in ss]
    np.testing.assert_array_equal(tau[0], tau[1])
    own = []                                                          # the union bound is at least every own bound
    for s in ss:
        s.union_bound(s.top, 1)
        own.append(s.tau.cpu().numpy())
    assert (tau[0] >= own[0]).all() and (tau[0] >= own[1]).all() and (tau[0] > np.minimum(own[0], own[1])).any()
    va = layout.views(rec_all)
    st_parts = va["status"].cpu().numpy()
    assert set(np.unique(st_parts)) <= {0, 2}
    assert st_parts[1, 2] == 2 and (va["I"][1, 2].cpu().numpy() == -1).all()     # second shard: nothing above the bound
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
    np.testing.assert_array_equal(best.cpu().numpy(), want["best"].cpu().numpy())
    assert I.cpu().numpy()[0, 0] == 11 and I.cpu().numpy()[0, 1] == n_rows // 2 + 7


def test_full_size_dump_properties():
    """BASELINE.json configs[1] at full size (170 M rows x 768 int8 = 130.6 GB, generated on the device): the CPU oracle
    cannot scan it, so parity rests on size-independent properties -- planted rows come back first with the score the
    host computes for them, every returned (id, score) pair re-computes on the host, rows sorted, ids unique, a random
    probe of other rows never beats the k-th score, the call is idempotent and certified, and cutting the dump into two
    shards and merging gives the same answer (partition invariance)."""
    import torch
    from densephrases_amd import Shard
    from densephrases_amd.synth import synthetic_rows
    free, _ = torch.cuda.mem_get_info(0)
    n, k, seed = 170_000_000, 10, 42
    if free < n * 768 + (8 << 30):
        pytest.skip("needs 140 GB of free HBM")
    rng = np.random.default_rng(2026)
    planted = rng.integers(0, n, 8)
    x = rng.normal(0, 0.5, (16, 768)).astype(np.float32)
    x[:8] = O.int8_to_float(np.stack([synthetic_rows(int(r), 1, seed)[0] for r in planted])) + \
        rng.normal(0, 0.1, (8, 768)).astype(np.float32)

    def host_scores(ids, qrow):
        rows = np.stack([synthetic_rows(int(i), 1, seed)[0] for i in ids])
        return O.int8_to_float(rows).astype(np.float64) @ x[qrow].astype(np.float64)

    s = Shard(n, device=0)
    s.fill_synthetic(seed=seed)
    s.finalize()
    D, I = s.search(x, k)
    assert s.stats()["uncertified"] == 0
    np.testing.assert_array_equal(I[:8, 0], planted)
    assert (np.diff(D, axis=1) <= 0).all() and ((I >= 0) & (I < n)).all()
    for r in range(16):
        assert len(set(I[r].tolist())) == k
        np.testing.assert_allclose(D[r], host_scores(I[r], r), rtol=2e-6, atol=1e-4)       # fp32 rounding of the score
        probe = rng.integers(0, n, 300)
        ps = host_scores(probe, r)
        beat = probe[ps > float(D[r, k - 1]) + 1e-3]
        assert set(beat.tolist()) <= set(I[r].tolist())
    D2, I2 = s.search(x, k)
    np.testing.assert_array_equal(I2, I)
    np.testing.assert_array_equal(D2, D)
    s.close()
    del s
    torch.cuda.empty_cache()

    h = (n // 2 // 800) * 800
    parts = []
    for lo, hi in ((0, h), (h, n)):
        p = Shard(hi - lo, device=0, id_base=lo)
        p.fill_synthetic(seed=seed)
        p.finalize()
        parts.append(p.search(x, k))
        assert p.stats()["uncertified"] == 0
        p.close()
        del p
    Dm = np.concatenate([parts[0][0], parts[1][0]], axis=1)
    Im = np.concatenate([parts[0][1], parts[1][1]], axis=1)
    for r in range(16):
        order = np.lexsort((Im[r], -Dm[r].astype(np.float64)))[:k]
        np.testing.assert_array_equal(Im[r][order], I[r])
        np.testing.assert_array_equal(Dm[r][order], D[r])


def test_mips_over_a_merged_index(tmp_path):
    """idx2id.hdf5 of a MERGED index (two sub-indexes, id offsets 0 and 10^8): MIPS loads the groups as dense rows with
    the id translation in libdph, the first-stage ids it reports are the reference's offset + local ids
    (index.py:135-140), and the answers equal the golden output of the reference's own index.py on the unsplit index
    (the split falls on a document boundary, so no window crosses it)."""
    import os
    import subprocess
    py39 = "/opt/conda/bin/python3.9"
    if not os.path.exists(py39):
        pytest.skip("no interpreter with h5py to write the fixture")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([py39, os.path.join(here, "_make_h5_dump.py"), os.path.join(here, "golden", "toy_dump.npz"),
                        str(tmp_path), "split:100000000"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("h5py writer failed: " + r.stderr[-200:])
    from densephrases_amd import MIPS
    idx_dir = os.path.join(str(tmp_path), "start", "toy_flat_none")
    mips = MIPS(phrase_dump_dir=os.path.join(str(tmp_path), "phrase"), index_path=os.path.join(idx_dir, "index.faiss"),
                idx2id_path=os.path.join(idx_dir, "idx2id.hdf5"), cuda=True)
    assert mips.index.ntotal == 261
    cut = int(mips.store.row_starts[1])
    for ci in (1, 3):
        c = CASES[ci]
        q = c["query_arr"].astype(np.float64)
        got = mips.search(q, q_texts=[f"q{i}" for i in range(c["B"])], top_k=c["top_k"], aggregate=c["aggregate"],
                          max_answer_length=c["L"], agg_strat=c["agg_strat"], return_sent=c["return_sent"],
                          return_idxs=c["return_idxs"])
        compare_results(got, c["results"], VECS, case=c)
    # first-stage ids are offset + local; get_idxs decodes them like the reference
    dense = mips.search_dense(CASES[1]["query_arr"], top_k=10)
    I = np.concatenate([dense[2], dense[5]]).reshape(-1)
    assert ((I < cut) | (I >= 100000000)).all() and (I >= 100000000).any() and (I < cut).any()
    rows = mips.store.rows_of_ids(I)
    np.testing.assert_array_equal(np.concatenate([dense[0], dense[3]]).reshape(-1), mips.store.row2doc[rows])
    np.testing.assert_array_equal(mips.shard.reconstruct(100000000), O.int8_to_float(mips.store.to_store().rows[cut]))


class _ThreadWorld:
    """torch.distributed stand-in for W ranks living in W threads of one process (one GPU): the two collectives the
    sharded search uses, implemented with a barrier and a shared slot list."""

    def __init__(self, world):
        import threading
        self.world, self.slots, self.bar = world, [None] * world, threading.Barrier(world)

    def rank_view(self, rank):
        import torch
        g = self

        class _Dist:
            @staticmethod
            def all_gather_into_tensor(out, inp):
                torch.cuda.synchronize()
                g.slots[rank] = inp
                g.bar.wait()
                o = out.view(g.world, -1)
                for r in range(g.world):
                    o[r].copy_(g.slots[r].reshape(-1))
                torch.cuda.synchronize()
                g.bar.wait()

            @staticmethod
            def all_reduce(t):
                torch.cuda.synchronize()
                g.slots[rank] = t.clone()
                g.bar.wait()
                acc = g.slots[0].clone()
                for r in range(1, g.world):
                    acc += g.slots[r]
                t.copy_(acc)
                torch.cuda.synchronize()
                g.bar.wait()

        return _Dist


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_mips_range_sharded_over_ranks_equals_single_rank(world):
    """MIPS(rank, world, dist): every rank loads only its document-aligned row range; search is a collective that
    returns, on every rank, exactly what the single-rank MIPS returns (dict for dict, incl. aggregation and the
    return_idxs vectors).  The ranks are threads of this process sharing the one GPU; world 4 and 8 are configs[3]'s and
    configs[2]'s rank counts.  One query row of the first batch is NON-FINITE: every rank flags it (status 3), the merge keeps the
    flag, nobody re-searches it, its result list is empty and every other query is answered as ever."""
    import threading
    from densephrases_amd import DocMeta, DocStore, MIPS
    from oracle.synth_dump import make_dump, make_queries
    docs = make_dump(seed=11, n_docs=300, d=768, n_par=4, words_per_par=(20, 40))
    conv = lambda ds: DocStore([DocMeta(m.doc_idx, m.title, m.context, m.f2o_start, m.word2char_start, m.word2char_end,  # noqa: E731
                                        m.start) for m in ds])
    single = MIPS.from_store(conv(docs))
    n = single.index.ntotal
    assert n > 20000
    rng = np.random.default_rng(4)
    q = make_queries(rng, single.store.rows, 12)
    texts = [f"q{i}" for i in range(12)]
    kw = dict(top_k=10, aggregate=True, agg_strat="opt1", max_answer_length=10)
    q[7, 11] = q[7, 768 + 11] = np.nan                   # (both halves: neither the start nor the end row of query 7 has candidates)
    want = single.search(q, q_texts=texts, **kw)
    assert want[7] == [] and all(len(w) > 0 for i, w in enumerate(want) if i != 7)
    want_vec = single.search(q[:4], q_texts=texts[:4], top_k=5, return_idxs=True)
    tw = _ThreadWorld(world)
    results, errors = [None] * world, []

    def run(rank):
        try:
            m = MIPS(None, "in-memory", None, device=0, _store=conv(docs), rank=rank, world=world, dist=tw.rank_view(rank))
            assert m.row_hi - m.row_lo < n and m.index.ntotal == n
            a = m.search(q, q_texts=texts, **kw)
            b = m.search(q[:4], q_texts=texts[:4], top_k=5, return_idxs=True)
            results[rank] = (a, b, (m.row_lo, m.row_hi))
        except Exception as e:                       # surface in the main thread; release the peers
            errors.append((rank, repr(e)))
            tw.bar.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    spans = sorted(r[2] for r in results)
    assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    for rank in range(world):
        a, b, _ = results[rank]
        for got, ref in ((a, want), (b, want_vec)):
            assert len(got) == len(ref)
            for g, w in zip(got, ref):
                assert len(g) == len(w)
                for x, y in zip(g, w):
                    for key in ("context", "title", "doc_idx", "start_pos", "end_pos", "start_idx", "end_idx", "answer"):
                        assert x[key] == y[key]
                    assert x["score"] == y["score"]
                    if y.get("start_vec") is not None:
                        np.testing.assert_array_equal(x["start_vec"], y["start_vec"])
                        np.testing.assert_array_equal(x["end_vec"], y["end_vec"])


def test_faiss_compat_speaks_the_protocol_index_py_uses(tmp_path):
    """densephrases_amd.faiss_compat as the ``faiss`` module: the exact call sequence of the reference's index.py
    (:30-34 read_index / downcast_index / chain.at(0).A / ntotal / d, :52-62 extract_index_ivf / nprobe / quantizer /
    index_cpu_to_all_gpus, :200 search, :286 reconstruct incl. the exception on unknown ids) over the reference-layout
    files, checked against the oracle."""
    import os
    import subprocess
    import sys
    py39 = "/opt/conda/bin/python3.9"
    if not os.path.exists(py39):
        pytest.skip("no interpreter with h5py to write the fixture")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([py39, os.path.join(here, "_make_h5_dump.py"), os.path.join(here, "golden", "toy_dump.npz"),
                        str(tmp_path)], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("h5py writer failed: " + r.stderr[-200:])
    import densephrases_amd.faiss_compat as fc
    old = sys.modules.get("faiss")
    try:
        faiss = fc.install()
        assert sys.modules["faiss"] is faiss
        index_path = os.path.join(str(tmp_path), "start", "toy_flat_none", "index.faiss")
        index = faiss.read_index(index_path, faiss.IO_FLAG_ONDISK_SAME_DIR)                      # index.py:30
        reconst_fn = faiss.downcast_index(index.index).reconstruct                               # :31
        R = faiss.vector_to_array(faiss.downcast_VectorTransform(index.chain.at(0)).A).reshape(index.d, index.d)   # :32
        np.testing.assert_array_equal(R, np.eye(768, dtype=np.float32))
        want = O.build_index_from_docs(load_toy_docs())
        assert index.ntotal == want.xb.shape[0] and index.d == 768                               # :34
        index_ivf = faiss.extract_index_ivf(index)                                                # :52-56
        index_ivf.nprobe = 256
        index_ivf.quantizer = faiss.index_cpu_to_all_gpus(index_ivf.quantizer)
        q = CASES[0]["query_arr"].astype(np.float32)
        stacked = np.concatenate(np.split(q, 2, axis=1), axis=0)
        D, I = index.search(stacked, 5)                                                           # :200
        Dr, Ir, D64 = _flat(stacked, want.xb, 5)
        ok, msg = O.topk_equivalent(D, I, D64, Ir)
        assert ok, msg
        np.testing.assert_array_equal(reconst_fn(7), O.int8_to_float(want.xb[7]))                 # :286
        with pytest.raises(RuntimeError):
            reconst_fn(want.xb.shape[0] + 5)                                                      # :285-288: caller substitutes zeros
    finally:
        if old is not None:
            sys.modules["faiss"] = old
        else:
            sys.modules.pop("faiss", None)


def test_fused_finest_ladder_level_reads_the_dump_once_and_changes_nothing():
    """Tuning key "ladder_fuse" (default on): the full scan skips the tiles the finest sampled level already scanned and
    accumulates into that level's buckets.  Same D / I / certificates as with the level re-scanned, at sizes with one, two
    and three ladder levels, a tile count that is not a multiple of the stride, 128 and 256 query rows, and duplicates of
    the best row INSIDE the fused tiles (tile 0, tile 32) and outside them."""
    from densephrases_amd import Shard
    for n_rows, n_q, k in ((600_011, 40, 10), (4_000_000, 256, 10), (40_000, 7, 100)):
        rng = np.random.default_rng(n_rows)
        xb = _rand_db(rng, n_rows)
        hot = xb[5].copy()
        for r in (5, 32 * 32 + 3, 32 * 64 + 31, 77, 1000, n_rows - 1):
            xb[r] = hot
        q = rng.normal(0, 0.5, (n_q, 768)).astype(np.float32)
        q[0] = hot.astype(np.float32) / 20 - 2
        s = Shard(n_rows, device=0)
        s.upload(xb)
        s.finalize()
        res = {}
        for fuse in (1, 0):
            s.set_tuning("ladder_fuse", fuse)
            D, I = s.search(q, k)
            st = s.stats()
            assert st["uncertified"] == 0
            assert st["fused_stride"] == 0 if fuse == 0 else (st["fused_stride"] >= 2 or n_rows < 100_000), st
            res[fuse] = (D, I, st)
        np.testing.assert_array_equal(res[1][1], res[0][1])
        np.testing.assert_array_equal(res[1][0], res[0][0])
        Dr, Ir, D64 = _flat(q, xb, k)
        ok, msg = O.topk_equivalent(res[1][0], res[1][1], D64, Ir)
        assert ok, msg
        assert set(res[1][1][0][:6].tolist()) == {5, 32 * 32 + 3, 32 * 64 + 31, 77, 1000, n_rows - 1}


@pytest.mark.parametrize("sched", [1, 2])
def test_staggered_hand_over_schedules_change_nothing(sched):
    """dph_scan.hip's hand-over schedules (tuning key ``scan_sched``: 1 = wave after wave, 2 = interleaved) move WHEN a wave
    stages its pieces of a tile, not what the scan computes: cold buckets hold every row once with its exact integer score
    (shards of one tile up to a few per workgroup, 128- and 256-row kernels), and on a shard long enough for the steady
    loop (dozens of tiles per workgroup and segment, several segments) the buckets under a bound and the search results
    are those of the lock-step schedule, key for key."""
    for n_rows, n_q in [(6000, 128), (700, 256), (33, 3), (5000, 200), (20000, 256)]:
        rng = np.random.default_rng(n_rows)
        xb = _rand_db(rng, n_rows)
        x = rng.normal(0, 0.5, (n_q, 768)).astype(np.float32)
        s = _shard(xb)
        s.set_tuning("scan_sched", sched)
        buckets, lost = s.debug_scan_buckets(x)
        assert not lost.any()
        q1, q2, _ = _host_digits(x)
        ref = 128 * (q1 @ xb.astype(np.int64).T) + q2 @ xb.astype(np.int64).T
        for q in range(0, n_q, 7):
            score, rows = buckets[q]
            assert np.array_equal(np.sort(rows), np.arange(n_rows)), f"{n_rows} rows, q{q}: rows missing / duplicated"
            np.testing.assert_array_equal(score.astype(np.int64), ref[q, rows.astype(np.int64)], err_msg=f"q{q}")
        s.close()
    from densephrases_amd import Shard
    n = 3_000_000
    s = Shard(n, device=0)
    s.fill_synthetic(seed=5, kind=0)
    s.finalize()
    rng = np.random.default_rng(sched)
    for n_q in (128, 256):
        x = rng.normal(0, 0.5, (n_q, 768)).astype(np.float32)
        # bounds a few dozen rows beat: the k-th best exact scores from a search of the same rows
        D, I = s.search(x, 40)
        q1, q2, sc = _host_digits(x)
        tau = np.floor((D[:, -1].astype(np.float64) + 2.0 * x.astype(np.float64).sum(1)) * 20.0 / sc).astype(np.int64) - 50
        tau = np.clip(tau, np.iinfo(np.int32).min + 1, np.iinfo(np.int32).max).astype(np.int32)
        got = {}
        for sch in (0, sched):
            s.set_tuning("scan_sched", sch)
            buckets, lost = s.debug_scan_buckets(x, tau=tau)
            assert not lost.any()
            got[sch] = [np.sort((b[0].astype(np.int64) << 32) | b[1].astype(np.int64)) for b in buckets]
        for q in range(n_q):
            assert got[0][q].size >= 30, f"q{q}: the bound admits {got[0][q].size} rows"
            np.testing.assert_array_equal(got[sched][q], got[0][q], err_msg=f"{n_q} rows, q{q}")
    x = rng.normal(0, 0.5, (300, 768)).astype(np.float32)
    s.set_tuning("scan_sched", 0)
    D0, I0 = s.search(x, 10)
    s.set_tuning("scan_sched", sched)
    D1, I1 = s.search(x, 10)
    np.testing.assert_array_equal(I1, I0)
    np.testing.assert_array_equal(D1, D0)
    assert s.stats()["uncertified"] == 0
    s.close()


def test_near_ties_at_the_kth_place_are_settled_on_the_bucket_without_a_second_scan():
    """300 exact copies of the best row: more candidates tie at the top than the first look re-scores (C = max(2k, k + 32)), so its
    certificate cannot close -- but every pair is in the row's bucket, and the second look at the SAME bucket with C = 2048 settles the
    row (``certified_reselect``): no re-scan of the shard.  (3000 copies: beyond that look too -- test_duplicate_rows_and_the_retry_chain.)"""
    rng = np.random.default_rng(6)
    n_rows = 400000
    xb = _rand_db(rng, n_rows)
    x = rng.normal(0, 0.5, (5, 768)).astype(np.float32)
    hot = xb[321].copy()
    x[2] = hot.astype(np.float32) / 20 - 2
    dup = rng.choice(np.arange(500, n_rows), 300, replace=False)
    xb[dup] = hot
    s = _shard(xb)
    s.profile_enable(True)
    D, I = s.search(x, 10)
    Dr, Ir, D64 = _flat(x, xb, 10)
    ok, msg = O.topk_equivalent(D, I, D64, Ir)
    assert ok, msg
    np.testing.assert_array_equal(I[2], Ir[2])     # exact ties: lowest ids first
    st = s.stats()
    assert st["uncertified"] == 0 and st["exact_fallback"] == 0
    assert st["certified_fast"] == 4 and st["certified_wide"] == 1 and st["certified_reselect"] == 1, st


def test_buckets_beyond_the_sort_capacity_are_cut_to_their_best_keys_not_rescanned():
    """A loose sampled bound (one ladder level at stride 8, the 1024-th best sampled score) leaves 8192 < keys <= 32768 in every
    bucket: more than the select step sorts.  Rounds 1-3 called that "lost pairs" and re-scanned the shard for such rows; now the best
    8192 keys by exact integer score go into the sort and the cut-off score enters the certificate like the bound of the scan: every
    row is certified by the first attempt, no retry, same answer as the oracle."""
    rng = np.random.default_rng(77)
    n_rows, n_q = 400000, 24
    xb = _rand_db(rng, n_rows)
    x = rng.normal(0, 0.5, (n_q, 768)).astype(np.float32)
    planted = rng.integers(0, n_rows, 8)
    x[:8] = (xb[planted].astype(np.float32) / 20 - 2 + rng.normal(0, 0.1, (8, 768))).astype(np.float32)
    s = _shard(xb)
    s.set_tuning("ladder", 8)
    s.set_tuning("sample_kp", 1024)
    s.set_tuning("retry_chain", 0)                  # whatever the first attempt leaves open would come back with status 1
    D, I = s.search(x, 10)
    raw, ov = s.debug_bucket_counts(n_q)
    assert (raw > 8192).all() and (raw <= 32768).all() and not ov.any(), raw
    st = s.stats()
    assert st["certified_fast"] == n_q and st["uncertified"] == 0, st
    Dr, Ir, D64 = _flat(x, xb, 10)
    ok, msg = O.topk_equivalent(D, I, D64, Ir)
    assert ok, msg
    np.testing.assert_array_equal(I[:8, 0], planted)
