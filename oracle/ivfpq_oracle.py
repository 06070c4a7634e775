"""CPU restatement of FAISS' IndexPreTransform(OPQMatrix) -> IndexIVFPQ (inner product, by_residual) search and
reconstruct -- the index the reference builds (build_phrase_index.py:108-116) and searches (index.py:30-33, 200, 286).
TEST INFRASTRUCTURE: only tests/, the golden generators and oracle/refshim import this; the product never does.

**Parity unpinned against FAISS**: faiss-gpu==1.6.5 (requirements.txt:2) is absent and not installable offline, the
reference ships no stored D/I arrays.  This restates the published algorithm:

    x'            = A x (+ b)                                  IndexPreTransform::apply_chain / LinearTransform::apply_noalloc
    coarse        = top-nprobe of <x', c_l>                    IndexFlatIP quantizer, index.py:53 nprobe = 256
    LUT[m][j]     = <x'_m, pq_centroid[m][j]>                  ProductQuantizer::compute_inner_prod_table
    score(code)   = <x', c_list> + sum_m LUT[m][code[m]]       IVFPQ scanner, METRIC_INNER_PRODUCT, by_residual
    top-k         = k largest scores, descending; -1 / -FLT_MAX padding when fewer than k codes were scanned
    reconstruct   = c_list + concat_m pq_centroid[m][code[m]]  IndexIVFPQ::reconstruct_from_offset (rotated space)

Arithmetic: FAISS computes all of this in fp32 with BLAS / SIMD summation orders it does not specify.  The restatement
fixes them so that an implementation can be held to it bit for bit: x', the coarse scores and the LUT entries are
accumulated in float64 and rounded once to fp32; a code's score is the fp32 sum `dis0 + LUT[0] + LUT[1] + ...` taken
SEQUENTIALLY in m (what FAISS' scalar scan loop does); ties are ordered (score desc, id asc), which FAISS leaves open.

Also here: a small trainer / adder (k-means coarse centroids, per-subspace k-means codebooks, a random rotation for the
OPQ matrix) to produce synthetic indexes in the reference's format -- FAISS' OPQ training itself is out of scope (offline
index build)."""
from __future__ import annotations

import numpy as np

from densephrases_amd.faiss_io import IVFPQIndex, LinearTransform, PreTransformIndex

NEG = np.float32(-3.4028234663852886e38)


def apply_chain(chain, x: np.ndarray, use_bias: bool = True) -> np.ndarray:
    """x [n, d_in] fp32 -> x' [n, d_out] fp32 (float64 accumulation, one rounding)."""
    x = np.asarray(x, np.float32)
    for t in chain:
        y = x.astype(np.float64) @ t.A.astype(np.float64).T
        if use_bias and t.b is not None:
            y = y + t.b.astype(np.float64)
        x = y.astype(np.float32)
    return x


def _ivf(index):
    return index.index if isinstance(index, PreTransformIndex) else index


def coarse_probe(xp: np.ndarray, centroids: np.ndarray, nprobe: int):
    """(lists [n, nprobe] int64 best first, dis0 [n, nprobe] fp32); ties (score desc, list asc)."""
    s = xp.astype(np.float64) @ centroids.astype(np.float64).T
    nprobe = min(nprobe, centroids.shape[0])
    order = np.lexsort((np.broadcast_to(np.arange(s.shape[1]), s.shape), -s), axis=1)[:, :nprobe]
    return order.astype(np.int64), np.take_along_axis(s, order, 1).astype(np.float32)


def lut(xp: np.ndarray, pq_centroids: np.ndarray) -> np.ndarray:
    """[n, M, ksub] fp32 inner-product tables"""
    M, ksub, dsub = pq_centroids.shape
    xs = xp.reshape(xp.shape[0], M, dsub).astype(np.float64)
    return np.einsum("nmt,mjt->nmj", xs, pq_centroids.astype(np.float64)).astype(np.float32)


def adc_scores(dis0: np.float32, table: np.ndarray, codes: np.ndarray) -> np.ndarray:
    """fp32 sequential sum dis0 + table[0][code[0]] + table[1][code[1]] + ... for codes [n, M]"""
    acc = np.full(codes.shape[0], dis0, np.float32)
    for m in range(codes.shape[1]):
        acc = (acc + table[m][codes[:, m]]).astype(np.float32)
    return acc


def search(index, x: np.ndarray, k: int, nprobe: int):
    """FAISS Index.search of the reference's index.py:200.  Returns D [n,k] fp32 (descending), I [n,k] int64."""
    ivf = _ivf(index)
    xp = apply_chain(index.chain, x) if isinstance(index, PreTransformIndex) else np.asarray(x, np.float32)
    n = xp.shape[0]
    lists, dis0 = coarse_probe(xp, ivf.centroids, nprobe)
    tables = lut(xp, ivf.pq_centroids)
    D = np.full((n, k), NEG, np.float32)
    I = np.full((n, k), -1, np.int64)
    for r in range(n):
        ss, ii = [], []
        for l, d0 in zip(lists[r], dis0[r]):
            codes = np.asarray(ivf.list_codes[l])
            if len(codes) == 0:
                continue
            base = d0 if ivf.by_residual else np.float32(0)
            ss.append(adc_scores(base, tables[r], codes))
            ii.append(np.asarray(ivf.list_ids[l]))
        if not ss:
            continue
        s, i = np.concatenate(ss), np.concatenate(ii)
        o = np.lexsort((i, -s.astype(np.float64)))[:k]
        D[r, :len(o)], I[r, :len(o)] = s[o], i[o]
    return D, I


class DirectMap:
    """id -> (list, offset): the Hashtable direct map make_direct_map builds (build_phrase_index.py:138-142)"""

    def __init__(self, ivf: IVFPQIndex):
        self.map = {}
        for l, ids in enumerate(ivf.list_ids):
            for o, i in enumerate(np.asarray(ids).tolist()):
                self.map[i] = (l, o)

    def get(self, i):
        return self.map.get(int(i))


def reconstruct(index, direct_map: DirectMap, i: int) -> np.ndarray:
    """faiss.downcast_index(index.index).reconstruct(i) (index.py:31): the decoded vector in the ROTATED space, fp32;
    raises like FAISS on an unknown id (the reference substitutes zeros, index.py:285-288)."""
    ivf = _ivf(index)
    lo = direct_map.get(i)
    if lo is None:
        raise RuntimeError(f"ivfpq oracle: id {i} not in the direct map")
    l, o = lo
    code = np.asarray(ivf.list_codes[l][o])
    v = np.concatenate([ivf.pq_centroids[m, code[m]] for m in range(ivf.M)]).astype(np.float32)
    if ivf.by_residual:
        v = (v + ivf.centroids[l]).astype(np.float32)
    return v


# ------------------------------------------------------------------------------------------------- building synthetic indexes
def _kmeans(x: np.ndarray, k: int, rng, iters: int = 8, ip: bool = False) -> np.ndarray:
    n = x.shape[0]
    c = x[rng.choice(n, k, replace=n < k)].astype(np.float64)
    if n < k:
        c = c + rng.normal(0, 1e-3, c.shape)
    x64 = x.astype(np.float64)
    for _ in range(iters):
        if ip:
            a = np.argmax(x64 @ c.T, 1)
        else:
            a = np.argmin((c * c).sum(1)[None, :] - 2.0 * (x64 @ c.T), 1)
        for j in range(k):
            m = a == j
            if m.any():
                c[j] = x64[m].mean(0)
    return c.astype(np.float32)


def random_rotation(d: int, rng) -> np.ndarray:
    q, r = np.linalg.qr(rng.normal(size=(d, d)))
    return (q * np.sign(np.diag(r))[None, :]).astype(np.float32)


def train(xb: np.ndarray, nlist: int, M: int, seed: int = 0, rotate: bool = True) -> PreTransformIndex:
    """An (empty) trained IndexPreTransform(OPQMatrix(d, M), IndexIVFPQ(IndexFlatIP, d, nlist, M, 8, IP)) over training
    vectors xb [n, d] fp32: the shape train_index gives its index (build_phrase_index.py:96-142)."""
    rng = np.random.default_rng(seed)
    xb = np.asarray(xb, np.float32)
    d = xb.shape[1]
    A = random_rotation(d, rng) if rotate else np.eye(d, dtype=np.float32)
    chain = [LinearTransform(A)]
    xp = apply_chain(chain, xb)
    cent = _kmeans(xp, nlist, rng, ip=True)
    a = np.argmax(xp.astype(np.float64) @ cent.astype(np.float64).T, 1)
    res = xp - cent[a]
    dsub = d // M
    pqc = np.stack([_kmeans(res[:, m * dsub:(m + 1) * dsub], 256, rng, iters=4) for m in range(M)])
    ivf = IVFPQIndex(d, nlist, M, 8, cent, pqc, [np.zeros((0, M), np.uint8) for _ in range(nlist)],
                     [np.zeros(0, np.int64) for _ in range(nlist)], True, 0, 1, 2)
    return PreTransformIndex(chain, ivf, d, True)


def add_with_ids(index: PreTransformIndex, xb: np.ndarray, ids: np.ndarray):
    """index.add_with_ids(xb, ids) (build_phrase_index.py:145-153): list = arg-max inner product over the centroids,
    code = nearest codeword (L2) of every sub-vector of the residual."""
    ivf = _ivf(index)
    xp = apply_chain(index.chain, np.asarray(xb, np.float32))
    a = np.argmax(xp.astype(np.float64) @ ivf.centroids.astype(np.float64).T, 1)
    res = (xp - ivf.centroids[a]).astype(np.float64)
    dsub = ivf.d // ivf.M
    codes = np.empty((len(xp), ivf.M), np.uint8)
    for m in range(ivf.M):
        c = ivf.pq_centroids[m].astype(np.float64)
        sub = res[:, m * dsub:(m + 1) * dsub]
        codes[:, m] = np.argmin((c * c).sum(1)[None, :] - 2.0 * (sub @ c.T), 1)
    ids = np.asarray(ids, np.int64)
    for l in np.unique(a):
        sel = a == l
        ivf.list_codes[l] = np.concatenate([np.asarray(ivf.list_codes[l]), codes[sel]])
        ivf.list_ids[l] = np.concatenate([np.asarray(ivf.list_ids[l]), ids[sel]])
    return index


def brute_force(index, direct_map: DirectMap, x: np.ndarray, k: int):
    """Second opinion sharing no code with ``search``: score of every stored id = <x', reconstruct(id)> in float64
    (the identity the ADC sum restates: <x', c + decode> = <x', c> + sum_m <x'_m, codeword_m>), over ALL lists."""
    ivf = _ivf(index)
    xp = apply_chain(index.chain, x).astype(np.float64)
    ids = np.concatenate([np.asarray(i) for i in ivf.list_ids])
    vecs = np.stack([reconstruct(index, direct_map, int(i)) for i in ids]).astype(np.float64)
    s = xp @ vecs.T
    o = np.lexsort((np.broadcast_to(ids, s.shape), -s), axis=1)[:, :k]
    return np.take_along_axis(s, o, 1), ids[o]
