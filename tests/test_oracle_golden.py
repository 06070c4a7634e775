"""The oracle's restatement of index.py, pinned against outputs of the REFERENCE's own code
(oracle/make_golden.py ran /root/reference/densephrases/index.py unmodified through oracle/refshim)."""
import numpy as np
import pytest

from oracle import mips_oracle as O
from tests._golden import compare_results, load_cases, load_toy_index

CASES, VECS = load_cases()


@pytest.fixture(scope="module")
def index():
    return load_toy_index()


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_search_dense_matches_reference(index, ci):
    c = CASES[ci]
    got = O.search_dense(index, c["query_arr"], c["top_k"])
    for a, b in zip(got, c["dense"]):
        b = np.asarray(b)
        if b.dtype.kind == "f":
            np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-5)
        else:
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("branch", ["ram", "hdf5"])
@pytest.mark.parametrize("ci", range(len(CASES)))
def test_search_matches_reference(index, ci, branch):
    """Both oracle window branches must reproduce the reference's results for BOTH reference branches
    (they only differ at masked window slots, which never win the argmax on an unfiltered index)."""
    c = CASES[ci]
    if c["return_idxs"] and branch != c["branch"]:
        pytest.skip("start_vec/end_vec of the candidate itself differ by reference branch (raw int8 vs float)")
    got = O.search(index, c["query_arr"].astype(np.float64), q_texts=None, top_k=c["top_k"],
                   aggregate=c["aggregate"], return_idxs=c["return_idxs"], max_answer_length=c["L"],
                   agg_strat=c["agg_strat"], return_sent=c["return_sent"], branch=branch)
    compare_results(got, c["results"], VECS, case=c)


def test_codec_roundtrip():
    x = np.linspace(-9, 5, 4001).astype(np.float32)
    n = O.float_to_int8(x)
    assert n.dtype == np.int8 and n.min() == -128 and n.max() == 127
    back = O.int8_to_float(n)
    inside = (x > -8.3) & (x < 4.3)
    assert np.abs(back[inside] - x[inside]).max() <= 0.025 + 1e-6
    lut = O.dequant_lut()
    assert lut.dtype == np.float32 and lut[128] == np.float32(-2.0) and lut[128 + 40] == np.float32(0.0)


def test_flat_ip_padding_and_ties():
    rng = np.random.default_rng(0)
    xb = O.float_to_int8(rng.normal(0, 0.6, (7, 16)).astype(np.float32))
    xb[5] = xb[2]                       # exact duplicate -> tie, lower id first
    q = rng.normal(0, 1, (3, 16)).astype(np.float32)
    D, I, D64 = O.flat_ip_search(q, xb, 10)
    assert (I[:, 7:] == -1).all() and (D[:, 7:] == -O.FLT_MAX).all()
    for r in range(3):
        pos2, pos5 = list(I[r]).index(2), list(I[r]).index(5)
        assert pos5 == pos2 + 1 and D64[r, pos2] == D64[r, pos5]
        assert (np.diff(D64[r, :7]) <= 0).all()
    ok, msg = O.topk_equivalent(D, I, D64, I)
    assert ok, msg


def test_flat_ip_blocked_equals_unblocked():
    rng = np.random.default_rng(1)
    xb = O.float_to_int8(rng.normal(0, 0.6, (3000, 32)).astype(np.float32))
    q = rng.normal(0, 1, (5, 32)).astype(np.float32)
    D1, I1, _ = O.flat_ip_search(q, xb, 10, block=256)
    D2, I2, _ = O.flat_ip_search(q, xb, 10, block=1 << 20)
    np.testing.assert_array_equal(I1, I2)
    np.testing.assert_array_equal(D1, D2)
    D3, I3 = O.flat_ip_search_sgemm(q, xb, 10)
    ok, msg = O.topk_equivalent(D3, I3, D2.astype(np.float64), I2, rtol=1e-5, atol=1e-4)
    assert ok, msg


def test_flat_ip_restatement_has_a_second_opinion_that_shares_no_code():
    """FAISS IndexFlatIP cannot be pinned offline: the restatement every golden rests on
    (``flat_ip_search``: float64 scores, canonical tie order) is held against a second one that shares no code with it
    and follows FAISS' own execution shape instead -- fp32 vectors, database blocks of 1024 rows, one sgemm per block,
    a running top-k that only admits strictly better scores (``flat_ip_search_fp32_resident``): the same ids except
    inside the fp32 noise band, on clustered data with duplicate rows."""
    rng = np.random.default_rng(31)
    n, k = 20000, 10
    centres = rng.normal(0, 0.5, (12, 768))
    xb = np.clip(np.rint(40 + 20 * (centres[rng.integers(0, 12, n)] + rng.normal(0, 0.3, (n, 768)))), -128, 127).astype(np.int8)
    xb[5000:5040] = xb[77]                                       # exact ties
    q = (O.int8_to_float(xb[rng.integers(0, n, 24)]) + rng.normal(0, 0.1, (24, 768))).astype(np.float32)
    D64, I64, S = O.flat_ip_search(q, xb, k)
    blocks = [O.int8_to_float(xb[r:r + 1024]) for r in range(0, n, 1024)]
    D32, I32 = O.flat_ip_search_fp32_resident(q, blocks, k)
    ok, msg = O.topk_equivalent(D32, I32, S, I64)
    assert ok, msg
    assert (I32 == I64).mean() > 0.97
