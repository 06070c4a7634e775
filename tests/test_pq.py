"""The reference's own index type -- IndexPreTransform(OPQMatrix) -> IndexIVFPQ, SURVEY.md section 8 rows a3 / f2 -- read
from FAISS 1.6 files (densephrases_amd/faiss_io.py), searched and reconstructed on the GPU (csrc/dph_pq.hip), held
against the restatement of FAISS' algorithm (oracle/ivfpq_oracle.py) and against goldens the reference's unmodified
index.py produced over such a file (oracle/make_golden_pq.py).  FAISS itself is absent offline: the file format and the
IVFPQ arithmetic are pinned to this repo's writer / oracle only (stated in DESIGN.md)."""
import json
import os
import pickle
import subprocess

import numpy as np
import pytest

from densephrases_amd import faiss_io as F
from oracle import ivfpq_oracle as P
from tests._golden import GOLD, compare_results, load_toy_docs

PY39 = "/opt/conda/bin/python3.9"
HERE = os.path.dirname(os.path.abspath(__file__))
INDEX_NAME = "toy_OPQ96_PQ"


def _pieces():
    return dict(np.load(os.path.join(GOLD, "pq_index.npz")))


def _golden_index():
    from oracle.make_golden_pq import build_index
    return build_index(_pieces())


def _pq_cases():
    cases = json.load(open(os.path.join(GOLD, "pq_cases.json")))
    z = np.load(os.path.join(GOLD, "pq_vecs.npz"))
    for c in cases:
        c["query_arr"] = z[f"query_{c['query']}"]
    return cases, z["vecs"]


def _random_index(seed, n, nlist, M=96, id_offset=0, dup=0):
    rng = np.random.default_rng(seed)
    centres = rng.normal(0, 0.5, (max(4, nlist // 2), 768)).astype(np.float32)
    xb = (centres[rng.integers(0, len(centres), n)] + rng.normal(0, 0.3, (n, 768))).astype(np.float32)
    if dup:
        xb[rng.choice(n, dup, replace=False)] = xb[0]               # identical vectors -> identical codes -> score ties
    ix = P.train(xb[: min(n, 3000)], nlist=nlist, M=M, seed=seed + 1)
    ids = np.arange(n, dtype=np.int64) + id_offset
    P.add_with_ids(ix, xb, ids)
    return ix, xb, rng


# ------------------------------------------------------------------------------------------------- CPU: format + oracle
def test_faiss_file_round_trip_array_and_ondisk_lists(tmp_path):
    ix = _golden_index()
    for ondisk in (False, True):
        p = str(tmp_path / f"index_{int(ondisk)}.faiss")
        F.write_index(ix, p, ondisk=ondisk)
        assert F.looks_like_faiss_index(p)
        back = F.read_index(p, F.IO_FLAG_ONDISK_SAME_DIR)
        assert isinstance(back, F.PreTransformIndex) and isinstance(back.index, F.IVFPQIndex)
        assert back.ntotal == ix.ntotal == 261 and back.index.nlist == 4 and back.index.M == 96 and back.index.by_residual
        np.testing.assert_array_equal(back.chain[0].A, ix.chain[0].A)
        np.testing.assert_array_equal(back.index.centroids, ix.index.centroids)
        np.testing.assert_array_equal(back.index.pq_centroids, ix.index.pq_centroids)
        for l in range(4):
            np.testing.assert_array_equal(np.asarray(back.index.list_codes[l]), ix.index.list_codes[l])
            np.testing.assert_array_equal(np.asarray(back.index.list_ids[l]), ix.index.list_ids[l])
        if ondisk:
            assert os.path.exists(str(tmp_path / "merged.invdata")) and back.index.ondisk_filename.endswith("merged.invdata")
    # the orthogonal OPQ stand-in really is orthogonal
    A = ix.chain[0].A.astype(np.float64)
    assert np.abs(A @ A.T - np.eye(768)).max() < 1e-5


def test_faiss_file_sparse_list_sizes_and_array_direct_map(tmp_path):
    """an index with most lists empty (FAISS then writes the sizes as (list, size) pairs, "sprs") and an ARRAY direct map
    (DirectMap type 1, dense ids) instead of the hash table"""
    rng = np.random.default_rng(2)
    nlist, M = 64, 96
    codes = [np.zeros((0, M), np.uint8)] * nlist
    ids = [np.zeros(0, np.int64)] * nlist
    codes[5], ids[5] = rng.integers(0, 256, (3, M), dtype=np.uint8), np.asarray([4, 0, 2], np.int64)
    codes[60], ids[60] = rng.integers(0, 256, (2, M), dtype=np.uint8), np.asarray([1, 3], np.int64)
    ivf = F.IVFPQIndex(768, nlist, M, 8, rng.normal(size=(nlist, 768)).astype(np.float32),
                       rng.normal(size=(M, 256, 8)).astype(np.float32), codes, ids, True, 0, 7, 1)
    p = str(tmp_path / "sparse.faiss")
    F.write_index(ivf, p)
    raw = open(p, "rb").read()
    assert b"sprs" in raw and b"full" not in raw
    back = F.read_index(p)
    assert isinstance(back, F.IVFPQIndex) and back.ntotal == 5 and back.nprobe == 7 and back.direct_map_type == 1
    for l in range(nlist):
        np.testing.assert_array_equal(np.asarray(back.list_codes[l]), codes[l])
        np.testing.assert_array_equal(np.asarray(back.list_ids[l]), ids[l])


def test_faiss_file_errors_are_reported_not_guessed(tmp_path):
    ix = _golden_index()
    p = str(tmp_path / "index.faiss")
    F.write_index(ix, p)
    raw = open(p, "rb").read()
    open(str(tmp_path / "cut.faiss"), "wb").write(raw[: len(raw) // 2])
    with pytest.raises(F.FaissFormatError):
        F.read_index(str(tmp_path / "cut.faiss"))
    open(str(tmp_path / "hnsw.faiss"), "wb").write(b"IHNp" + raw[4:])
    with pytest.raises(F.FaissFormatError):
        F.read_index(str(tmp_path / "hnsw.faiss"))
    assert not F.looks_like_faiss_index(str(tmp_path / "missing.faiss"))
    flat = F.FlatIndex(768, np.zeros((3, 768), np.float32))
    F.write_index(flat, str(tmp_path / "flat.faiss"))
    assert F.read_index(str(tmp_path / "flat.faiss")).ntotal == 3


@pytest.mark.parametrize("ondisk", [False, True])
def test_damaged_faiss_files_are_refused_not_read_into_the_heap(tmp_path, ondisk):
    """Truncations and flipped bytes (length fields, list tables, the name of the inverted-list file): the reader either still
    parses the file (a flip inside bulk data) or raises FaissFormatError -- never MemoryError from a damaged length, a numpy
    reshape error, a codec error or a missing-file error from a damaged name."""
    rng = np.random.default_rng(0)
    d, nlist, M = 768, 8, 48
    sizes = [5, 0, 3, 7, 0, 1, 2, 4]
    ids, o = [], 0
    for n in sizes:
        ids.append(np.arange(o, o + n, dtype=np.int64))
        o += n
    ix = F.PreTransformIndex([F.LinearTransform(np.eye(d, dtype=np.float32))],
                             F.IVFPQIndex(d, nlist, M, 8, rng.normal(size=(nlist, d)).astype(np.float32),
                                          rng.normal(size=(M, 256, d // M)).astype(np.float32),
                                          [rng.integers(0, 256, (n, M), dtype=np.uint8) for n in sizes], ids), d)
    good = str(tmp_path / "index.faiss")
    F.write_index(ix, good, ondisk=ondisk)
    raw = open(good, "rb").read()
    bad = str(tmp_path / "damaged.faiss")
    outcomes = {"ok": 0, "refused": 0}
    for it in range(900):
        b = bytearray(raw)
        if it % 3 == 0:
            b = b[:int(rng.integers(0, len(b)))]
        else:
            for _ in range(int(rng.integers(1, 4))):
                pos = int(rng.choice([rng.integers(0, 200), rng.integers(len(b) - 4000, len(b)), rng.integers(0, len(b))]))
                b[pos] = int(rng.integers(0, 256))
        with open(bad, "wb") as f:
            f.write(bytes(b))
        try:
            F.read_index(bad, F.IO_FLAG_ONDISK_SAME_DIR)
            outcomes["ok"] += 1
        except F.FaissFormatError:
            outcomes["refused"] += 1
    assert outcomes["refused"] > 300 and outcomes["ok"] > 100, outcomes


def test_oracle_adc_search_agrees_with_brute_force_over_reconstructions():
    """two restatements that share no scoring code: the ADC sum (LUT adds) and <x', reconstruct(id)> over ALL lists"""
    ix, xb, rng = _random_index(3, 1500, 8)
    dm = P.DirectMap(ix.index)
    q = (xb[:7] + rng.normal(0, 0.05, (7, 768))).astype(np.float32)
    D, I = P.search(ix, q, 10, nprobe=8)
    S, J = P.brute_force(ix, dm, q, 10)
    np.testing.assert_array_equal(I, J)
    np.testing.assert_allclose(D, S, rtol=2e-6, atol=1e-4)
    with pytest.raises(RuntimeError):
        P.reconstruct(ix, dm, 10 ** 12)


# ------------------------------------------------------------------------------------------------- GPU: kernels vs oracle
def _shard(ix):
    from densephrases_amd import Shard
    s = Shard.from_faiss_index(ix, device=0)
    n = s.ntotal
    s.set_idx2id(np.zeros(n, np.int32), np.arange(n, dtype=np.int32))
    s.set_f2o(np.zeros(1, np.int32), np.asarray([0, n], np.int64), np.arange(n, dtype=np.int32))
    s.finalize()
    return s


def _same_topk(D, I, Dr, Ir):
    """ids equal, except swaps / substitutions between scores closer than the fp32 resolution of the two implementations"""
    np.testing.assert_allclose(D, Dr, rtol=3e-7, atol=1e-5)
    bad = np.nonzero(I != Ir)
    for r, c in zip(*bad):
        assert abs(float(D[r, c]) - float(Dr[r, c])) <= 3e-7 * abs(float(Dr[r, c])) + 1e-5, (r, c, I[r, c], Ir[r, c])
        assert I[r, c] in Ir[r] or abs(float(D[r, c]) - float(Dr[r, -1])) <= 3e-7 * abs(float(Dr[r, -1])) + 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("n,nlist,nprobe,k,nq", [(6000, 32, 8, 10, 9), (6000, 32, 32, 1, 3), (6000, 32, 1, 100, 5),
                                                 (40000, 4, 4, 10, 4), (300, 16, 16, 1000, 2), (2000, 16, 5, 20, 300)])
def test_pq_search_matches_the_ivfpq_oracle(n, nlist, nprobe, k, nq):
    """lists of several 8192-code segments (40000 / 4), k larger than what the probed lists hold (padding), more than 256
    query rows, nprobe 1 .. nlist"""
    ix, xb, rng = _random_index(11 + n + nprobe, n, nlist)
    s = _shard(ix)
    q = (xb[rng.integers(0, n, nq)] + rng.normal(0, 0.1, (nq, 768))).astype(np.float32)
    D, I = s.search_ivf(q, k, nprobe)
    Dr, Ir = P.search(ix, q, k, nprobe)
    _same_topk(D, I, Dr, Ir)
    assert s.stats()["uncertified"] == 0
    np.testing.assert_array_equal(s.transform(), ix.chain[0].A)


@pytest.mark.gpu
def test_pq_ids_beyond_32_bits_duplicates_and_reconstruct():
    """ids of a merged index (offset 5e9), 40 identical vectors (equal codes, equal scores: ordered by id), reconstruct
    bit for bit = centroid + decoded residual in the rotated space, unknown ids raise like FAISS"""
    from densephrases_amd._lib import DphError
    ix, xb, rng = _random_index(5, 3000, 8, id_offset=5_000_000_000, dup=40)
    s = _shard(ix)
    q = np.concatenate([xb[:1], xb[100:104]]).astype(np.float32)
    D, I = s.search_ivf(q, 50, 8)
    Dr, Ir = P.search(ix, q, 50, 8)
    _same_topk(D, I, Dr, Ir)
    assert I.min() >= 5_000_000_000
    top = I[0][D[0] == D[0, 0]]
    assert len(top) >= 41 and (np.diff(top) > 0).all()             # the tie block comes back in id order
    dm = P.DirectMap(ix.index)
    for i in (5_000_000_000, 5_000_000_017, 5_000_002_999):
        np.testing.assert_array_equal(s.reconstruct(i), P.reconstruct(ix, dm, i))
    for bad in (17, 5_000_003_000, -1):
        with pytest.raises(DphError):
            s.reconstruct(bad)
    # the default entry point searches with nprobe 256 (index.py:53,62), i.e. every list of this index
    D2, I2 = s.search(q, 50)
    Dr2, Ir2 = P.search(ix, q, 50, 256)
    _same_topk(D2, I2, Dr2, Ir2)


@pytest.mark.gpu
@pytest.mark.parametrize("M,by_residual,bias", [(64, True, False), (128, True, True), (96, False, False), (48, True, True)])
def test_pq_variants_sub_quantiser_counts_bias_and_non_residual_codes(M, by_residual, bias):
    """M = 64 / 128 / 48 (12-, 6-, 16-dimensional sub-vectors; M = 128 halves the scan segment), a pre-transform WITH a
    bias vector (x' = A x + b: the search uses it, the window un-rotation does not), codes of the vectors themselves
    instead of residuals (by_residual = False: no <x', centroid> term), empty lists, nprobe beyond the non-empty lists"""
    rng = np.random.default_rng(100 + M)
    n, nlist = 5000, 24
    centres = rng.normal(0, 0.5, (6, 768)).astype(np.float32)
    xb = (centres[rng.integers(0, 6, n)] + rng.normal(0, 0.3, (n, 768))).astype(np.float32)
    ix = P.train(xb[:2500], nlist=nlist, M=M, seed=M)
    if bias:
        ix.chain[0].b = rng.normal(0, 0.05, 768).astype(np.float32)
    ix.index.by_residual = by_residual
    P.add_with_ids(ix, xb, np.arange(n, dtype=np.int64) * 3 + 7)          # sparse ids
    for l in (3, 11):                                                     # two lists emptied after the fact
        ix.index.list_codes[l] = np.zeros((0, M), np.uint8)
        ix.index.list_ids[l] = np.zeros(0, np.int64)
    s = _shard(ix)
    q = (xb[rng.integers(0, n, 6)] + rng.normal(0, 0.1, (6, 768))).astype(np.float32)
    for nprobe, k in ((nlist, 10), (5, 3), (1, 1024)):
        D, I = s.search_ivf(q, k, nprobe)
        Dr, Ir = P.search(ix, q, k, nprobe)
        _same_topk(D, I, Dr, Ir)
    dm = P.DirectMap(ix.index)
    some = np.asarray(ix.index.list_ids[0][:3]).tolist() + np.asarray(ix.index.list_ids[nlist - 1][-2:]).tolist()
    for i in some:
        np.testing.assert_array_equal(s.reconstruct(int(i)), P.reconstruct(ix, dm, int(i)))


@pytest.mark.gpu
def test_pq_with_65536_lists_goes_through_the_bf16x3_coarse_gemm_and_the_one_pass_select():
    """The long-quantizer path (the reference's index has 2^20 lists): coarse scores from three bf16 MFMAs per product
    (hi*hi + hi*lo + lo*hi) with the float64 re-rank band widened by the split's error, the nprobe-th score found in one
    pass over the scores (sampled estimate + candidates in LDS), the work grouped by query row (probe lists, one LUT load
    per 64 lists).  Duplicate centroids put exact ties on the probe boundary: the probed set must still be the float64
    oracle's (score desc, list id asc)."""
    rng = np.random.default_rng(65)
    nlist, M, n = 65536, 96, 6000
    cent = rng.normal(0, 0.5, (nlist, 768)).astype(np.float32)
    cent[40000:40064] = cent[123]                                   # 64 copies of one centroid
    pqc = rng.normal(0, 0.1, (M, 256, 768 // M)).astype(np.float32)
    lists = rng.integers(0, nlist, n)
    lists[:400] = rng.choice(np.concatenate([[123], np.arange(40000, 40064)]), 400)     # codes inside the tie block
    codes = rng.integers(0, 256, (n, M), dtype=np.uint8)
    order = np.argsort(lists, kind="stable")
    ids = np.arange(n, dtype=np.int64)
    list_codes = [np.zeros((0, M), np.uint8)] * nlist
    list_ids = [np.zeros(0, np.int64)] * nlist
    ls, cs, is_ = lists[order], codes[order], ids[order]
    cuts = np.nonzero(np.diff(ls))[0] + 1
    for seg_l, seg_c, seg_i in zip(np.split(ls, cuts), np.split(cs, cuts), np.split(is_, cuts)):
        list_codes[int(seg_l[0])], list_ids[int(seg_l[0])] = seg_c, seg_i
    A = P.random_rotation(768, rng)
    ix = F.PreTransformIndex([F.LinearTransform(A)], F.IVFPQIndex(768, nlist, M, 8, cent, pqc, list_codes, list_ids, True, 0, 1, 2), 768, True)
    s = _shard(ix)
    q = rng.normal(0, 0.5, (5, 768)).astype(np.float32)
    q[0] = (A.T @ cent[123]).astype(np.float32)                      # x' = A q = the duplicated centroid: the tie block is on top
    for nprobe, k in ((256, 10), (40, 10), (1, 5)):
        Dr, Ir = P.search(ix, q, k, nprobe)
        for filt in (5, 4, 3, 2, 1, 0):
            # 5 (round 5): the filter as a SCAN -- the centroids as 24 KiB pieces with an int8 tile's byte layout through the flat scan's
            # feed (dph_scan.hip MODE 3), hits out of the pair pool;
            # 3 / 2 / 1: the one-product filter GEMM with the threshold test in its epilogue (round 4; 1 = centroids and queries staged
            # through LDS, 2 = the same with the centroid stream loaded non-temporal, 3 = centroids straight into MFMA operand registers
            # from a fragment-ordered image, 4 = the same with every workgroup on a contiguous run of tiles), 0: the three-product chain alone
            s.set_tuning("coarse_filter", filt)
            D, I = s.search_ivf(q, k, nprobe)
            _same_topk(D, I, Dr, Ir)
            if filt:
                failed_over, emitted = s.debug_pq_coarse()
                assert failed_over is False and emitted >= nprobe * q.shape[0], (nprobe, failed_over, emitted)


@pytest.mark.gpu
def test_pq_filter_scan_over_several_groups_of_128_query_rows():
    """A pass of 300 query rows: the filter scan reads the centroid image once per group of 128 rows (128 + 128 + 44) and every
    group's hits are dealt into the per-row candidate lists straight from the scan's chunks (dph_coarse_bucket_chunks_kernel, query
    base = the group's first row).  Same top-k as the float64 oracle, no fail-over, and the same as the GEMM form."""
    rng = np.random.default_rng(67)
    nlist, M, n = 65536, 96, 9000
    cent = rng.normal(0, 0.5, (nlist, 768)).astype(np.float32)
    pqc = rng.normal(0, 0.1, (M, 256, 768 // M)).astype(np.float32)
    lists = rng.integers(0, nlist, n)
    codes = rng.integers(0, 256, (n, M), dtype=np.uint8)
    order = np.argsort(lists, kind="stable")
    ids = np.arange(n, dtype=np.int64)
    list_codes = [np.zeros((0, M), np.uint8)] * nlist
    list_ids = [np.zeros(0, np.int64)] * nlist
    ls, cs, is_ = lists[order], codes[order], ids[order]
    cuts = np.nonzero(np.diff(ls))[0] + 1
    for seg_l, seg_c, seg_i in zip(np.split(ls, cuts), np.split(cs, cuts), np.split(is_, cuts)):
        list_codes[int(seg_l[0])], list_ids[int(seg_l[0])] = seg_c, seg_i
    A = P.random_rotation(768, rng)
    ix = F.PreTransformIndex([F.LinearTransform(A)], F.IVFPQIndex(768, nlist, M, 8, cent, pqc, list_codes, list_ids, True, 0, 1, 2), 768, True)
    s = _shard(ix)
    q = rng.normal(0, 0.5, (300, 768)).astype(np.float32)
    Dr, Ir = P.search(ix, q, 10, 256)
    for filt in (5, 3):
        s.set_tuning("coarse_filter", filt)
        D, I = s.search_ivf(q, 10, 256)
        _same_topk(D, I, Dr, Ir)
        failed_over, emitted = s.debug_pq_coarse()
        assert failed_over is False and emitted >= 256 * q.shape[0], (filt, failed_over, emitted)
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_q", [129, 1024])
def test_pq_filter_scan_teams_at_the_edges_of_the_group_count(n_q):
    """The teams launch (dph_scan.hip MODE 4) with 2 groups of which the second holds ONE row, and with the full pass of 1024 rows
    (8 groups, teams of 8 workgroups): every row's probe set is the float64 oracle's, no fail-over; with tuning key coarse_teams = 0
    (one launch per 128 rows) the same answer."""
    rng = np.random.default_rng(69 + n_q)
    nlist, M = 65536, 96
    cent = rng.normal(0, 0.5, (nlist, 768)).astype(np.float32)
    ix, A = _index_from_list_numbers(rng, nlist, M, rng.integers(0, nlist, 12000), cent)
    s = _shard(ix)
    q = rng.normal(0, 0.5, (n_q, 768)).astype(np.float32)
    Dr, Ir = P.search(ix, q, 5, 16)
    for teams in (1, 0):
        s.set_tuning("coarse_teams", teams)
        D, I = s.search_ivf(q, 5, 16)
        _same_topk(D, I, Dr, Ir)
        failed_over, emitted = s.debug_pq_coarse()
        assert failed_over is False and emitted >= 16 * n_q, (teams, failed_over, emitted)
    s.close()


def test_sampled_segment_bound_never_drops_a_member_of_the_top_k():
    """pq_segment_finish<true> in numpy (dph_pq.hip): the bound of a segment of n > 2048 keys is the k-th largest of the strided
    sample keys[i * n // 1024], i < 1024 (positions over the whole segment) -- the k-th largest of a SUBSET of the row's scores, hence never above the row's true k-th:
    whatever order the segments of a row arrive in and whatever bound the row already has, the union of what they append contains
    the exact top-k, and the row's bound only rises.  (CPU: the rule, not the kernel.)"""
    rng = np.random.default_rng(5)
    PQ_THREADS = 1024
    for trial in range(20):
        k = int(rng.choice([1, 10, 100, 128]))
        segs = [rng.normal(0, 1, int(n)).astype(np.float32) for n in rng.integers(1, 12288, rng.integers(1, 9))]
        if trial % 3 == 0:
            segs[0][: len(segs[0]) // 2] = segs[0][0]                      # ties
        if trial % 4 == 1:
            segs[-1] = np.sort(segs[-1])                                   # the worst order for a strided sample: ascending
        bound, appended = -np.inf, []
        for keys in segs:
            n = len(keys)
            ns = PQ_THREADS if (n > 2 * PQ_THREADS and k <= PQ_THREADS // 8) else n
            sample = keys[(np.arange(ns) * n) // ns]
            T, b0 = bound, bound
            if int((sample >= b0).sum()) >= k:
                kth = np.sort(sample)[-k]
                assert kth >= b0
                T = max(T, kth)
                bound = max(bound, kth)
            if int((sample >= b0).sum()) > 0 or ns < n:
                appended.append(keys[keys >= T])
        allk = np.concatenate(segs)
        got = np.concatenate(appended) if appended else np.zeros(0, np.float32)
        kk = min(k, len(allk))
        true_top = np.sort(allk)[-kk:]
        np.testing.assert_array_equal(np.sort(got)[-kk:], true_top)       # nothing of the top-k was dropped
        assert bound <= true_top[0] or len(allk) < k                       # the bound stayed a lower bound of the true k-th


def test_sampled_segment_bound_appends_within_the_reserved_capacity():
    """What a COLD segment (row bound still -inf) appends under the sampled rule, against what pq_ensure reserves for it
    (2 k seg / 1024 per segment + 512 per row).  A segment is a sequence of lists, each with its own <x', centroid> offset: the row's
    best list at the tail, at the head, keys ascending.  The sample positions floor(i n / 1024) span the whole segment; with
    i * floor(n / 1024) (rounds 5-6) up to 1023 keys at the end were never sampled and a best list there was appended whole -- the
    overflow tests/test_fuzz_gpu.py found (CPU: the rule, not the kernel)."""
    rng = np.random.default_rng(6)
    worst_new, worst_old = 0, 0
    for trial in range(60):
        k = int(rng.choice([1, 2, 10, 128]))
        n = int(rng.integers(2049, 12289))
        best = int(rng.integers(300, 1000))
        rest = rng.normal(0, 1, n - best).astype(np.float32)
        top = (rng.normal(0, 1, best) + 50.0).astype(np.float32)              # one list far above the others
        keys = [np.concatenate([rest, top]), np.concatenate([top, rest]), np.sort(np.concatenate([rest, top]))][trial % 3]
        sample = keys[(np.arange(1024) * n) // 1024]
        count = int((keys >= np.sort(sample)[-k]).sum())
        reserved = 2 * k * (12288 // 1024) + 512
        assert count <= reserved, (trial, k, n, count, reserved)
        worst_new = max(worst_new, count - 2 * k * 12)
        old = keys[np.arange(1024) * (n // 1024)]
        worst_old = max(worst_old, int((keys >= np.sort(old)[-k]).sum()) - 2 * k * 12)
    assert worst_new < 200 and worst_old > 512                                # (the old positions did cross the reservation)


@pytest.mark.gpu
def test_pq_row_major_units_of_many_segments_take_their_bound_from_a_sample():
    """Row-major scan (many short lists on average) whose units are LONG: 64 neighbouring lists of 3000 codes each -- 192 k codes in the
    unit a query near them takes, sixteen segments of 12288 -- so the segment bound comes from the strided 1024-key sample
    (pq_segment_finish<true>), the segments append more than k keys each and pq_final_kernel selects among a few thousand.  Exact
    top-k all the same, k = 10 and k = 100; k = 200 is past the sample's reach and takes the exact k-th of every segment."""
    rng = np.random.default_rng(68)
    nlist, M = 65536, 96
    heavy, per = 64, 3000
    base = rng.normal(0, 0.5, 768).astype(np.float32)
    cent = rng.normal(0, 0.5, (nlist, 768)).astype(np.float32)
    cent[:heavy] = base + rng.normal(0, 0.01, (heavy, 768)).astype(np.float32)
    pqc = rng.normal(0, 0.1, (M, 256, 768 // M)).astype(np.float32)
    n_light = 9000
    lists = np.concatenate([np.repeat(np.arange(heavy), per), rng.integers(heavy, nlist, n_light)])
    n = len(lists)
    codes = rng.integers(0, 256, (n, M), dtype=np.uint8)
    order = np.argsort(lists, kind="stable")
    ids = rng.permutation(n).astype(np.int64)
    list_codes = [np.zeros((0, M), np.uint8)] * nlist
    list_ids = [np.zeros(0, np.int64)] * nlist
    ls, cs, is_ = lists[order], codes[order], ids[order]
    cuts = np.nonzero(np.diff(ls))[0] + 1
    for seg_l, seg_c, seg_i in zip(np.split(ls, cuts), np.split(cs, cuts), np.split(is_, cuts)):
        list_codes[int(seg_l[0])], list_ids[int(seg_l[0])] = seg_c, seg_i
    A = P.random_rotation(768, rng)
    ix = F.PreTransformIndex([F.LinearTransform(A)], F.IVFPQIndex(768, nlist, M, 8, cent, pqc, list_codes, list_ids, True, 0, 1, 2), 768, True)
    s = _shard(ix)
    q = np.stack([(A.T @ (base * f)).astype(np.float32) for f in (1.0, 0.7, 1.3)] + [rng.normal(0, 0.5, 768).astype(np.float32)])
    for k, nprobe in ((10, 64), (100, 256), (200, 64)):
        Dr, Ir = P.search(ix, q, k, nprobe)
        D, I = s.search_ivf(q, k, nprobe)
        _same_topk(D, I, Dr, Ir)
        assert s.stats()["uncertified"] == 0
    s.close()


def _index_from_list_numbers(rng, nlist, M, lists, cent, ids=None):
    """an IndexPreTransform(OPQ) -> IndexIVFPQ whose code i (random bytes) lies in list lists[i]"""
    n = len(lists)
    pqc = rng.normal(0, 0.1, (M, 256, 768 // M)).astype(np.float32)
    codes = rng.integers(0, 256, (n, M), dtype=np.uint8)
    order = np.argsort(lists, kind="stable")
    ids = np.arange(n, dtype=np.int64) if ids is None else ids
    list_codes = [np.zeros((0, M), np.uint8)] * nlist
    list_ids = [np.zeros(0, np.int64)] * nlist
    ls, cs, is_ = lists[order], codes[order], ids[order]
    cuts = np.nonzero(np.diff(ls))[0] + 1
    for seg_l, seg_c, seg_i in zip(np.split(ls, cuts), np.split(cs, cuts), np.split(is_, cuts)):
        if len(seg_l):
            list_codes[int(seg_l[0])], list_ids[int(seg_l[0])] = seg_c, seg_i
    A = P.random_rotation(768, rng)
    ix = F.PreTransformIndex([F.LinearTransform(A)], F.IVFPQIndex(768, nlist, M, 8, cent, pqc, list_codes, list_ids, True, 0, 1, 2), 768, True)
    return ix, A


@pytest.mark.gpu
def test_pq_one_list_of_600k_codes_among_65536_short_ones_is_cut_by_code_count():
    """What a k-means quantizer over token vectors leaves (build_phrase_index.py:113-116,156-279): ONE list 4000 x the mean, and it is
    probed.  The row-major scan cuts its work by code count (pq_units_kernel: units of <= 12288 codes of a group's 64 lists), so the
    600 k codes of list 4242 are 49 units of 49 workgroups -- each with its own sampled bound, all appending to the row's candidates.
    Same top-k as the oracle for rows that probe the giant list first, last (its centroid on the probe boundary), and not at all."""
    rng = np.random.default_rng(71)
    nlist, M, giant, n_giant = 65536, 96, 4242, 600_000
    cent = rng.normal(0, 0.5, (nlist, 768)).astype(np.float32)
    lists = np.concatenate([np.full(n_giant, giant), rng.integers(0, nlist, 20000)])
    ix, A = _index_from_list_numbers(rng, nlist, M, lists, cent, ids=rng.permutation(len(lists)).astype(np.int64))
    s = _shard(ix)
    q = rng.normal(0, 0.5, (6, 768)).astype(np.float32)
    q[0] = (A.T @ cent[giant]).astype(np.float32)                         # the giant list is this row's best list
    q[1] = (A.T @ (cent[giant] * 0.5 + cent[9] * 0.5)).astype(np.float32)
    q[2] = (A.T @ (cent[giant] * 0.16)).astype(np.float32) + q[2]         # ... somewhere inside the probe set
    for k, nprobe in ((10, 256), (100, 64), (1, 1)):
        Dr, Ir = P.search(ix, q, k, nprobe)
        D, I = s.search_ivf(q, k, nprobe)
        _same_topk(D, I, Dr, Ir)
        assert s.stats()["uncertified"] == 0
    lists_probed, _ = P.coarse_probe(P.apply_chain(ix.chain, q), cent, 256)
    assert (lists_probed[:3] == giant).any(1).all() and not (lists_probed[3:] == giant).any()      # both kinds of row were tested
    s.close()


@pytest.mark.gpu
def test_pq_coarse_filter_with_a_few_long_centroids_does_not_fail_over():
    """Centroids of very different lengths (an index trained on token vectors: the lists of frequent tokens are long AND their
    centroids are): 40 of 65536 centroids are 1.6 .. 2.4 x the others.  The one-product filter's error bound is per list -- heavy
    candidates get their exact float64 score before the band is cut (dph_coarse_select_kernel) -- so the band keeps the width the
    ordinary centroids give it and the pass does NOT fail over to the bf16x3 chain; the probed set is the float64 oracle's.  With
    one centroid 12 x the others the pool test cannot hold any more and the pass fails over: same answer, the slow way."""
    rng = np.random.default_rng(73)
    nlist, M = 65536, 96
    cent = rng.normal(0, 0.5, (nlist, 768)).astype(np.float32)
    heavy = rng.choice(nlist, 40, replace=False)
    cent[heavy] *= rng.uniform(1.6, 2.4, (40, 1)).astype(np.float32)
    lists = np.concatenate([rng.integers(0, nlist, 9000), rng.choice(heavy, 3000)])
    ix, A = _index_from_list_numbers(rng, nlist, M, lists, cent)
    s = _shard(ix)
    q = rng.normal(0, 0.5, (9, 768)).astype(np.float32)
    q[0] = (A.T @ cent[heavy[0]]).astype(np.float32) * 0.5
    for nprobe, k in ((256, 10), (32, 10), (1, 3)):
        Dr, Ir = P.search(ix, q, k, nprobe)
        D, I = s.search_ivf(q, k, nprobe)
        _same_topk(D, I, Dr, Ir)
        failed_over, emitted = s.debug_pq_coarse()
        assert failed_over is False and emitted >= nprobe * q.shape[0], (nprobe, failed_over, emitted)
    s.close()
    cent[heavy[1]] *= 6.0
    ix, A = _index_from_list_numbers(rng, nlist, M, lists, cent)
    s = _shard(ix)
    Dr, Ir = P.search(ix, q, 10, 256)
    D, I = s.search_ivf(q, 10, 256)
    _same_topk(D, I, Dr, Ir)
    assert s.debug_pq_coarse()[0] is True
    s.close()


@pytest.mark.gpu
def test_pq_list_major_scan_cuts_a_long_list_into_chunks():
    """Long lists on average (mean >= 2048 codes: the list-major scan) with one of 400 k codes: its (list, row) pairs are cut into
    chunks of four segments (pq_pairs_kernel), the chunks of a pair go to different workgroups; exact top-k all the same."""
    rng = np.random.default_rng(72)
    nlist, M = 32, 96
    cent = rng.normal(0, 0.5, (nlist, 768)).astype(np.float32)
    lists = np.concatenate([np.full(400_000, 5), rng.integers(0, nlist, 70_000)])
    ix, A = _index_from_list_numbers(rng, nlist, M, lists, cent)
    s = _shard(ix)
    q = rng.normal(0, 0.5, (5, 768)).astype(np.float32)
    q[0] = (A.T @ cent[5]).astype(np.float32)
    for k, nprobe in ((10, 8), (200, 32), (1, 1)):
        Dr, Ir = P.search(ix, q, k, nprobe)
        D, I = s.search_ivf(q, k, nprobe)
        _same_topk(D, I, Dr, Ir)
        assert s.stats()["uncertified"] == 0
    s.close()


@pytest.mark.gpu
def test_pq_coarse_filter_fails_over_when_the_error_band_overflows():
    """A 2^16-list quantizer with 1500 copies of one centroid.  nprobe 6000: the one-product filter would need more candidates per row
    than it keeps (and its error band around the 6000-th score holds more lists than the float64 re-rank takes), so the pass fails
    over -- on the device -- to the three-product chain with its 10^3 times narrower band: same probe set as the float64 oracle.
    nprobe 256 cuts THROUGH the tie block: the sampled threshold lands on the tie value itself, the filter cannot place its band below
    it and fails over; the chain's re-rank holds the 1501 exact ties (< 2048) and takes the lowest list ids.  A tie block beyond the
    re-rank's capacity is reported, not answered from a truncated band: DPH_E_UNCERTIFIED."""
    from densephrases_amd._lib import DphError
    rng = np.random.default_rng(66)
    nlist, M, n = 65536, 96, 8000

    def build(n_dup):
        cent = rng.normal(0, 0.5, (nlist, 768)).astype(np.float32)
        cent[30000:30000 + n_dup] = cent[77]
        pqc = rng.normal(0, 0.1, (M, 256, 768 // M)).astype(np.float32)
        lists = rng.integers(0, nlist, n)
        lists[:3000] = rng.choice(np.concatenate([[77], np.arange(30000, 30000 + n_dup)]), 3000)
        codes = rng.integers(0, 256, (n, M), dtype=np.uint8)
        order = np.argsort(lists, kind="stable")
        ids = np.arange(n, dtype=np.int64)
        list_codes = [np.zeros((0, M), np.uint8)] * nlist
        list_ids = [np.zeros(0, np.int64)] * nlist
        ls, cs, is_ = lists[order], codes[order], ids[order]
        cuts = np.nonzero(np.diff(ls))[0] + 1
        for seg_l, seg_c, seg_i in zip(np.split(ls, cuts), np.split(cs, cuts), np.split(is_, cuts)):
            list_codes[int(seg_l[0])], list_ids[int(seg_l[0])] = seg_c, seg_i
        A = P.random_rotation(768, rng)
        ix = F.PreTransformIndex([F.LinearTransform(A)], F.IVFPQIndex(768, nlist, M, 8, cent, pqc, list_codes, list_ids, True, 0, 1, 2), 768, True)
        q = rng.normal(0, 0.5, (3, 768)).astype(np.float32)
        q[0] = (A.T @ cent[77]).astype(np.float32)
        return ix, q

    ix, q = build(1500)
    s = _shard(ix)
    for nprobe in (6000, 256):
        Dr, Ir = P.search(ix, q, 10, nprobe)
        D, I = s.search_ivf(q, 10, nprobe)
        failed_over, _ = s.debug_pq_coarse()
        assert failed_over is True, nprobe
        _same_topk(D, I, Dr, Ir)
    s.close()
    ix, q = build(3000)
    s = _shard(ix)
    with pytest.raises(DphError) as e:
        s.search_ivf(q, 10, 256)
    assert e.value.code == -6
    D, I = s.search_ivf(q[1:], 10, 256)                  # the rows that do not sit on the tie block are answered as ever
    Dr, Ir = P.search(ix, q[1:], 10, 256)
    _same_topk(D, I, Dr, Ir)
    s.close()


# ------------------------------------------------------------------------------------------------- GPU: MIPS over a real index file
def _write_pq_layout(root):
    """the reference's dump_dir with REAL files: phrase/0-1.hdf5 + idx2id.hdf5 (h5py of python3.9), blosc meta_compressed.pkl,
    start/<name>/index.faiss + merged.invdata written by faiss_io from the committed index pieces"""
    from tests.test_reference_callers import _blosc_compress
    if not os.path.exists(PY39):
        pytest.skip("no interpreter with h5py to write the fixture")
    os.makedirs(root, exist_ok=True)
    r = subprocess.run([PY39, os.path.join(HERE, "_make_h5_dump.py"), os.path.join(GOLD, "toy_dump.npz"), root,
                        f"name:{INDEX_NAME}"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("h5py writer failed: " + r.stderr[-300:])
    meta = {str(m.doc_idx): {"word2char_start": _blosc_compress(m.word2char_start.tobytes()),
                             "word2char_end": _blosc_compress(m.word2char_end.tobytes()),
                             "f2o_start": _blosc_compress(m.f2o_start.tobytes()),
                             "context": _blosc_compress(m.context.encode("utf-8")), "title": m.title,
                             "dtypes": {"word2char_start": m.word2char_start.dtype, "word2char_end": m.word2char_end.dtype,
                                        "f2o_start": m.f2o_start.dtype}} for m in load_toy_docs()}
    with open(os.path.join(root, "meta_compressed.pkl"), "wb") as f:
        pickle.dump(meta, f)
    F.write_index(_golden_index(), os.path.join(root, "start", INDEX_NAME, "index.faiss"), ondisk=True)
    return root


@pytest.mark.gpu
def test_mips_over_a_real_opq_ivfpq_file_matches_the_reference_goldens(tmp_path):
    """densephrases_amd.MIPS(index_path = a FAISS file with OPQ + IVFPQ) against what the reference's own index.py returned
    over the same file: dense ids / scores, windows over reconstructed vectors un-rotated by R, aggregation, return_sent,
    and the return_idxs vectors with the reference's doubly rotated pred_*_vecs"""
    from densephrases_amd import MIPS
    root = _write_pq_layout(str(tmp_path / "dump"))
    idx_dir = os.path.join(root, "start", INDEX_NAME)
    mips = MIPS(phrase_dump_dir=os.path.join(root, "phrase"), index_path=os.path.join(idx_dir, "index.faiss"),
                idx2id_path=os.path.join(idx_dir, "idx2id.hdf5"), cuda=True)
    assert mips.index.ntotal == 261 and mips.max_idx == int(1e9)
    np.testing.assert_array_equal(mips.R, _golden_index().chain[0].A)
    cases, vecs = _pq_cases()
    for c in cases:
        q = c["query_arr"]
        dense = mips.search_dense(q, q_texts=None, top_k=c["top_k"])
        for got, want in zip(dense, c["dense"]):
            want = np.asarray(want)
            if want.dtype.kind == "f":
                np.testing.assert_allclose(np.asarray(got), want, rtol=3e-7, atol=1e-5)
            else:
                np.testing.assert_array_equal(np.asarray(got), want)
        got = mips.search(q.astype(np.float64), q_texts=[f"q{i}" for i in range(c["B"])], top_k=c["top_k"],
                          aggregate=c["aggregate"], return_idxs=c["return_idxs"], max_answer_length=c["L"],
                          agg_strat=c["agg_strat"], return_sent=c["return_sent"])
        # window scores: the reference sums fp32 products of the un-rotated vector (torch), the kernel rounds an exact
        # <A q, v'> once: 1e-6 relative; the vectors go through one / two fp32 768-term rotations: 1e-5 absolute
        _compare_pq(got, c["results"], vecs)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 4])
def test_mips_over_a_pq_file_range_sharded_over_ranks_equals_single_rank(tmp_path, world):
    """north_star: "(or PQ-compressed) phrase dump ... range-partitioned across the GPUs".  W ranks (threads sharing the GPU, the
    collectives of tests/test_gpu_search.py's _ThreadWorld) each load the OPQ matrix, all centroids and codebooks and the CODES of their
    own row range of the same index.faiss; every rank probes the same lists, scores its share, the records merge: the result dicts
    -- dense scores, windows over reconstructed vectors, aggregation, return_idxs vectors -- are those of one rank holding everything."""
    import threading
    from densephrases_amd import MIPS
    from tests.test_gpu_search import _ThreadWorld
    root = _write_pq_layout(str(tmp_path / "dump"))
    idx_dir = os.path.join(root, "start", INDEX_NAME)
    paths = dict(phrase_dump_dir=os.path.join(root, "phrase"), index_path=os.path.join(idx_dir, "index.faiss"),
                 idx2id_path=os.path.join(idx_dir, "idx2id.hdf5"), cuda=True)
    single = MIPS(**paths)
    cases, _ = _pq_cases()
    want = []
    for c in cases:
        want.append(single.search(c["query_arr"].astype(np.float64), q_texts=[f"q{i}" for i in range(c["B"])], top_k=c["top_k"],
                                  aggregate=c["aggregate"], return_idxs=c["return_idxs"], max_answer_length=c["L"],
                                  agg_strat=c["agg_strat"], return_sent=c["return_sent"]))
    tw = _ThreadWorld(world)
    results, errors = [None] * world, []

    def run(rank):
        try:
            m = MIPS(rank=rank, world=world, dist=tw.rank_view(rank), device=0, **paths)
            assert m.shard.n_rows == m.row_hi - m.row_lo < 261 and m.index.ntotal == 261
            out = []
            for c in cases:
                out.append(m.search(c["query_arr"].astype(np.float64), q_texts=[f"q{i}" for i in range(c["B"])], top_k=c["top_k"],
                                    aggregate=c["aggregate"], return_idxs=c["return_idxs"], max_answer_length=c["L"],
                                    agg_strat=c["agg_strat"], return_sent=c["return_sent"]))
            results[rank] = (out, (m.row_lo, m.row_hi))
        except Exception as e:                       # surface in the main thread; release the peers
            errors.append((rank, repr(e)))
            tw.bar.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    spans = sorted(r[1] for r in results)
    assert spans[0][0] == 0 and spans[-1][1] == 261 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    for rank in range(world):
        for got, ref in zip(results[rank][0], want):
            assert len(got) == len(ref)
            for g, w in zip(got, ref):
                assert len(g) == len(w)
                for x, y in zip(g, w):
                    for key in ("context", "title", "doc_idx", "start_pos", "end_pos", "start_idx", "end_idx", "answer"):
                        assert x[key] == y[key], (key, x[key], y[key])
                    assert x["score"] == y["score"]
                    if y.get("start_vec") is not None:
                        np.testing.assert_array_equal(x["start_vec"], y["start_vec"])
                        np.testing.assert_array_equal(x["end_vec"], y["end_vec"])


def _compare_pq(got, want, vecs):
    assert len(got) == len(want)
    for qi, (g, w) in enumerate(zip(got, want)):
        assert len(g) == len(w), f"query {qi}: {len(g)} results vs reference {len(w)}"
        for ri, (a, b) in enumerate(zip(g, w)):
            for key in ("context", "title", "doc_idx", "start_pos", "end_pos", "start_idx", "end_idx", "answer"):
                assert a[key] == b[key], f"query {qi} result {ri} field {key}: {a[key]!r} != {b[key]!r}"
            assert np.isclose(a["score"], b["score"], rtol=2e-6, atol=1e-4), (qi, ri, a["score"], b["score"])
            for key in ("start_vec", "end_vec"):
                if b.get(key) is None:
                    assert a.get(key) is None
                else:
                    np.testing.assert_allclose(np.asarray(a[key], np.float32), vecs[b[key]], rtol=1e-5, atol=2e-5)


@pytest.mark.gpu
def test_reference_index_py_runs_unmodified_over_the_pq_index_in_hbm(tmp_path):
    """the reference's densephrases/index.py, executed unmodified with densephrases_amd.faiss_compat as its `faiss`: its
    reconstruct loops (index.py:282-300), `@ R` (:340,365) and `.dot(R)` (:381-389) run over libdph's IVFPQ search /
    reconstruct / OPQ matrix; results = the goldens the same file produced over the oracle's FAISS stand-in"""
    from oracle import refshim
    if not refshim.reference_available():
        pytest.skip("reference byte code not built (oracle/build_ref.py)")
    from densephrases_amd import faiss_compat
    from oracle.refshim import real_io
    root = _write_pq_layout(str(tmp_path / "dump"))
    idx_dir = os.path.join(root, "start", INDEX_NAME)
    ref = refshim.install(faiss_module=faiss_compat, h5py_module=real_io.h5py_module(), blosc_module=real_io.blosc_module())
    mips = ref.MIPS(phrase_dump_dir=os.path.join(root, "phrase"), index_path=os.path.join(idx_dir, "index.faiss"),
                    idx2id_path=os.path.join(idx_dir, "idx2id.hdf5"), cuda=False)
    cases, vecs = _pq_cases()
    for c in cases:
        q = c["query_arr"]
        got = mips.search(q.astype(np.float64), q_texts=[f"q{i}" for i in range(c["B"])], top_k=c["top_k"],
                          aggregate=c["aggregate"], return_idxs=c["return_idxs"], max_answer_length=c["L"],
                          agg_strat=c["agg_strat"], return_sent=c["return_sent"])
        _compare_pq(got, c["results"], vecs)


def test_faiss_file_round_trip_property(tmp_path):
    """random small indexes (list counts, sub-quantiser counts, empty lists, bias, non-residual, all three direct-map types,
    both list containers) survive writer -> reader unchanged, and the oracle returns the same answers from the re-read index"""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=int(__import__('os').environ.get('DPH_HYP_EXAMPLES', '12')), deadline=None, derandomize=True, database=None)
    @given(seed=st.integers(0, 10 ** 6), nlist=st.sampled_from([1, 3, 8]), M=st.sampled_from([16, 48, 96]),
           ondisk=st.booleans(), dm=st.sampled_from([0, 1, 2]), bias=st.booleans(), by_res=st.booleans())
    def run(seed, nlist, M, ondisk, dm, bias, by_res):
        rng = np.random.default_rng(seed)
        n = int(rng.integers(0, 40))
        lists = rng.integers(0, nlist, n)
        ids = rng.permutation(n).astype(np.int64) if dm == 1 else rng.choice(10 ** 12, n, replace=False).astype(np.int64)
        codes = rng.integers(0, 256, (n, M), dtype=np.uint8)
        ivf = F.IVFPQIndex(768, nlist, M, 8, rng.normal(size=(nlist, 768)).astype(np.float32),
                           rng.normal(size=(M, 256, 768 // M)).astype(np.float32),
                           [codes[lists == l] for l in range(nlist)], [ids[lists == l] for l in range(nlist)], by_res, 0, 5, dm)
        A = rng.normal(size=(768, 768)).astype(np.float32)
        ix = F.PreTransformIndex([F.LinearTransform(A, rng.normal(size=768).astype(np.float32) if bias else None)], ivf, 768, True)
        d = tmp_path / f"case_{seed}_{nlist}_{M}_{int(ondisk)}_{dm}"
        d.mkdir(exist_ok=True)
        p = str(d / "index.faiss")
        F.write_index(ix, p, ondisk=ondisk)
        back = F.read_index(p, F.IO_FLAG_ONDISK_SAME_DIR)
        assert back.ntotal == n and back.index.by_residual == by_res and back.index.direct_map_type == dm and back.index.nprobe == 5
        np.testing.assert_array_equal(back.chain[0].A, A)
        assert (back.chain[0].b is None) == (not bias)
        for l in range(nlist):
            np.testing.assert_array_equal(np.asarray(back.index.list_codes[l]), ivf.list_codes[l])
            np.testing.assert_array_equal(np.asarray(back.index.list_ids[l]), ivf.list_ids[l])
        if n:
            q = rng.normal(size=(2, 768)).astype(np.float32)
            D0, I0 = P.search(ix, q, 5, nprobe=nlist)
            D1, I1 = P.search(back, q, 5, nprobe=nlist)
            np.testing.assert_array_equal(I0, I1)
            np.testing.assert_array_equal(D0, D1)

    run()
