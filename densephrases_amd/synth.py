"""Host replica of libdph's on-device synthetic dump generators (dph_index_fill_synthetic_kind, csrc/dph_quant.hip
``dph_fill_kernel``), integer arithmetic only (Irwin-Hall sum of four hashed bytes) so the GPU and this numpy code
agree bit for bit:
  kind 0 -- BASELINE.md config 2: rows i.i.d. ~ float_to_int8(N(0, 0.6^2), -2, 20) = 40 + 12 z;
  kind 1 -- SURVEY.md 8(d) config 4: mixture of 4096 Gaussians (n = 40 + 10 z_cluster + 5 z_row) in which every row
            with hash(row) % 999983 == 0 is a saturated outlier (+127 / -128 by a per-row 16-bit sign pattern).
  kind 3 -- kind 1 without the saturated rows (SURVEY 8(d) config 4 data as specified; IVF recall is measured on it);
  kind 2 -- a document-ordered dump: runs of 56..200 consecutive near-duplicate rows (n = 40 + 11.5 z_run + 3.4 z_row;
            the runs are the two parts of every block of 256 rows, split at 56 + hash(block) % 145).
  kind 4 -- an anisotropic dump shaped like BERT-family phrase vectors: kind 2's runs, every row scaled by a log-normal
            factor (per run x per row, 2^(e/16); ~2 % of the rows at twice the median norm) and five rogue dimensions
            (ROGUE_DIMS) whose code sits near ROGUE_MEANS for every row.
Used by the tests and by bench.py (planted queries, bounded CPU sample)."""
from __future__ import annotations

import numpy as np

DIM = 768
ROGUE_DIMS = (77, 138, 381, 588, 729)
ROGUE_MEANS = (-105, 110, -95, 100, -110)
_POW2_16TH = np.array([4096, 4277, 4467, 4664, 4871, 5087, 5312, 5547, 5793, 6049, 6317, 6597, 6889, 7194, 7512, 7845], dtype=np.int64)


def _hash32(lo: np.ndarray, hi: np.ndarray, seed: int) -> np.ndarray:
    m = np.uint64(0xFFFFFFFF)
    h = ((lo * np.uint64(0x9E3779B1)) & m) ^ ((hi * np.uint64(0x85EBCA77) + np.uint64(seed & 0xFFFFFFFF)) & m)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x7FEB352D)) & m
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x846CA68B)) & m
    h ^= h >> np.uint64(16)
    return h


def _ih4(h: np.ndarray) -> np.ndarray:
    return ((h & np.uint64(255)) + ((h >> np.uint64(8)) & np.uint64(255)) + ((h >> np.uint64(16)) & np.uint64(255))
            + (h >> np.uint64(24))).astype(np.int64) - 510


def synthetic_rows(row0: int, n: int, seed: int = 42, kind: int = 0) -> np.ndarray:
    """int8 [n, 768]: rows row0 .. row0+n-1 of the synthetic dump (global row index = id_base + local row)."""
    m32 = np.uint64(0xFFFFFFFF)
    seed_lo, seed_hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    e = (np.arange(n * DIM, dtype=np.uint64) + np.uint64(row0 * DIM))
    lo = e & m32
    hi = (e >> np.uint64(32)) ^ np.uint64(seed_hi)
    h = _hash32(lo, hi, seed)
    if kind == 0:
        v = 40 + ((_ih4(h) * 5321 + 32768) >> 16)
        return np.clip(v, -128, 127).astype(np.int8).reshape(n, DIM)
    rows = np.arange(n, dtype=np.uint64) + np.uint64(row0)
    if kind == 4:
        j = np.arange(DIM, dtype=np.uint64)
        run = synthetic_run_of_row(rows, seed)
        hc = _hash32((run[:, None] * np.uint64(768) + j[None, :]) & m32,
                     np.broadcast_to((np.uint64(0xD7) + (run >> np.uint64(20)))[:, None], (n, DIM)), (seed_lo + 0x51ED) & 0xFFFFFFFF)
        er = (_ih4(_hash32(run, np.uint64(0xA5) + (run >> np.uint64(20)), (seed_lo + 0x7A11) & 0xFFFFFFFF)) * 2560 + 32768) >> 16
        ew = (_ih4(_hash32(rows & m32, (rows >> np.uint64(32)) ^ np.uint64(0x1B873593), seed_lo ^ seed_hi ^ 0x3C6E)) * 2560 + 32768) >> 16
        e16 = er + ew
        ip = e16 >> 4
        base = _POW2_16TH[e16 & 15]
        s12 = np.where(ip >= 0, base << np.maximum(ip, 0), base >> np.maximum(-ip, 0))[:, None]
        hh = _ih4(h).reshape(n, DIM)
        v = 40 + (((((_ih4(hc) * 5099 + 32768) >> 16) + ((hh * 1508 + 32768) >> 16)) * s12) >> 12)
        for d, m in zip(ROGUE_DIMS, ROGUE_MEANS):
            v[:, d] = m + ((((hh[:, d] * 2661 + 32768) >> 16) * s12[:, 0]) >> 12)
        return np.clip(v, -128, 127).astype(np.int8)
    if kind == 2:
        j = np.arange(DIM, dtype=np.uint64)
        run = synthetic_run_of_row(rows, seed)
        hc = _hash32((run[:, None] * np.uint64(768) + j[None, :]) & m32,
                     np.broadcast_to((np.uint64(0xD7) + (run >> np.uint64(20)))[:, None], (n, DIM)), (seed_lo + 0x51ED) & 0xFFFFFFFF)
        v = 40 + ((_ih4(hc) * 5099 + 32768) >> 16) + ((_ih4(h).reshape(n, DIM) * 1508 + 32768) >> 16)
        return np.clip(v, -128, 127).astype(np.int8)
    hr = _hash32(rows & m32, (rows >> np.uint64(32)) ^ np.uint64(0x5BD1E995), seed_lo ^ seed_hi)      # [n]
    cluster = hr & np.uint64(4095)
    j = np.arange(DIM, dtype=np.uint64)
    hc = _hash32((cluster[:, None] * np.uint64(768) + j[None, :]) & m32, np.full((n, DIM), 0xC1, dtype=np.uint64),
                 (seed_lo + 0x9E37) & 0xFFFFFFFF)
    v = 40 + ((_ih4(hc) * 4434 + 32768) >> 16) + ((_ih4(h).reshape(n, DIM) * 2217 + 32768) >> 16)
    outlier = ((hr % np.uint64(999983)) == 0) & (kind == 1)
    if outlier.any():
        sign = ((hr[:, None] >> (j[None, :] & np.uint64(15))) & np.uint64(1)).astype(bool)
        v = np.where(outlier[:, None], np.where(sign, 127, -128), v)
    return np.clip(v, -128, 127).astype(np.int8)


def synthetic_run_of_row(rows: np.ndarray, seed: int = 42) -> np.ndarray:
    """run number (uint64, the 32-bit value the device computes) of global row indices of the kind-2 dump"""
    m32 = np.uint64(0xFFFFFFFF)
    rows = np.asarray(rows, dtype=np.uint64)
    block = rows >> np.uint64(8)
    split = np.uint64(56) + _hash32(block & m32, (block >> np.uint64(32)) ^ np.uint64(0x2545F491),
                                    (seed & 0xFFFFFFFF) ^ ((seed >> 32) & 0xFFFFFFFF)) % np.uint64(145)
    return (np.uint64(2) * (block & m32) + ((rows & np.uint64(255)) >= split).astype(np.uint64)) & m32


def synthetic_outlier_rows(n_total: int, seed: int = 42, limit: int = 1 << 22):
    """global indices < min(n_total, limit) of the saturated rows of the kind-1 dump (a scan over hashes; tests only)"""
    m32 = np.uint64(0xFFFFFFFF)
    rows = np.arange(min(n_total, limit), dtype=np.uint64)
    hr = _hash32(rows & m32, (rows >> np.uint64(32)) ^ np.uint64(0x5BD1E995), (seed & 0xFFFFFFFF) ^ ((seed >> 32) & 0xFFFFFFFF))
    return np.nonzero((hr % np.uint64(999983)) == 0)[0]


def synthetic_queries(n: int, seed: int = 1234, planted_rows=None, noise: float = 0.1) -> np.ndarray:
    """fp32 [n, 768] query rows.  With ``planted_rows`` (int8 [n,768]) every query is a de-quantised stored row
    plus N(0, noise^2): the nearest neighbour is known."""
    rng = np.random.default_rng(seed)
    if planted_rows is not None:
        base = planted_rows.astype(np.float32) / 20.0 - 2.0
        return (base + rng.normal(0.0, noise, base.shape)).astype(np.float32)
    return rng.normal(0.0, 0.6, (n, DIM)).astype(np.float32)


WORDS = "alpha bravo charlie delta echo foxtrot golf hotel india juliet kilo lima mike november oscar papa".split()


class SynthDocStore:
    """Metadata of the synthetic dumps bench.py / tools/e2e_mips.py search end to end: document d = rows
    [100 d, 100 d + 100): 100 tokens in 4 paragraphs, every token kept (f2o = identity).  Implements the ``doc_meta``
    protocol MIPS.from_shard takes."""

    def __init__(self, cache_docs: int = 200000):
        self._cache, self._cap = {}, cache_docs

    def doc_meta(self, d):
        from .dump import DocMeta
        m = self._cache.get(d)
        if m is None:
            rng = np.random.default_rng(d)
            toks = [WORDS[i] for i in rng.integers(0, len(WORDS), 100)]
            starts, ends, parts, pos = [], [], [], 0
            for p in range(4):
                par = toks[25 * p:25 * p + 25]
                for i, w in enumerate(par):
                    starts.append(pos)
                    ends.append(pos + len(w))
                    pos += len(w) + (1 if i < 24 else 0)
                parts.append(" ".join(par))
                pos += len(" [PAR] ")
            m = DocMeta(d, f"Doc {d}", " [PAR] ".join(parts), np.arange(100, dtype=np.int64),
                        np.asarray(starts, np.int32), np.asarray(ends, np.int32))
            if len(self._cache) < self._cap:
                self._cache[d] = m
        return m


def _pq_list_sizes(rng, n_codes: int, nlist: int, skew: str = "") -> np.ndarray:
    """list sizes: weights x n, floored.  ``skew`` "" -- exponential weights (the first draw of the generator's stream): near-uniform,
    the longest list ~14 x the mean; "zipf" -- weight of the list of rank r = r^-0.7 in a random order (170 M codes in 2^20 lists:
    the longest list ~800 k codes, ten above 150 k, a hundred above 30 k); "lognormal" -- exp(1.5 z) (a handful near 10^5);
    "giant" -- exponential weights and ONE list (number 4242 mod nlist) of 600 k codes (or n / 4)."""
    if skew == "zipf":
        w = (np.arange(1, nlist + 1, dtype=np.float64) ** -0.7)[rng.permutation(nlist)]
    elif skew == "lognormal":
        w = np.exp(1.5 * rng.normal(0.0, 1.0, nlist))
    elif skew in ("", "giant"):
        w = rng.exponential(1.0, nlist)
    else:
        raise ValueError(f"unknown PQ list-size skew {skew!r}")
    giant = min(600_000, n_codes // 4) if skew == "giant" else 0
    sizes = np.floor(w / w.sum() * (n_codes - giant)).astype(np.int64)
    # ... and the codes the flooring left over one each to the first lists (putting them all into list 0 made ONE list of ~nlist / 2
    # codes -- half a million at 2^20 lists, 3000 x the mean -- by accident; "giant" asks for it)
    rem = n_codes - giant - int(sizes.sum())
    sizes[: rem % nlist] += 1
    sizes += rem // nlist
    if giant:
        sizes[4242 % nlist] += giant
    return sizes


def synthetic_pq_list_sizes(n_codes: int, nlist: int, seed: int = 0, skew: str = "") -> np.ndarray:
    """the list sizes of ``synthetic_pq_parts(n_codes, nlist, ., seed)`` alone (no centroids, codebooks or codes are drawn)"""
    return _pq_list_sizes(np.random.default_rng(seed), n_codes, nlist, skew)


def synthetic_pq_parts(n_codes: int, nlist: int, M: int = 96, seed: int = 0, skew: str = "", hot: float = 0.18):
    """The pieces of the synthetic PQ index (below), reproducible from the seed alone: list sizes, OPQ matrix, coarse centroids,
    codebooks and the 2^20-code block every megacode of the index is a rolled copy of -- code of position p =
    block[(p % 2^20 - (p >> 20) % 97) mod 2^20], id of position p = p.
    With a ``skew`` the long lists are also the most PROBED ones, as in an index trained on token vectors (the lists of frequent tokens
    are long and near many queries): the centroid of a list of s > 16 x mean codes is scaled by 1 + hot * (log2(s / mean) - 4), at
    most 2.5 -- under inner product a longer centroid enters more probe sets (x 2.4: ~7 % of random queries probe it at nprobe 256 of
    2^20; lists below 16 x the mean -- every list of the un-skewed index -- keep their centroid)."""
    import math
    rng = np.random.default_rng(seed)
    sizes = _pq_list_sizes(rng, n_codes, nlist, skew)
    v = rng.normal(0, 1, 768).astype(np.float32)
    vv = np.float32(math.fsum(float(x) * float(x) for x in v))
    A = np.ascontiguousarray((np.eye(768, dtype=np.float32) - (np.float32(2.0) / vv) * np.outer(v, v).astype(np.float32))[rng.permutation(768)])
    cent = rng.normal(0, 0.5, (nlist, 768)).astype(np.float32)
    if skew and hot > 0:
        scale = np.clip(1.0 + hot * (np.log2(np.maximum(sizes, 1) / (n_codes / nlist)) - 4.0), 1.0, 2.5).astype(np.float32)
        cent *= scale[:, None]
    pqc = np.ascontiguousarray(rng.normal(0, 0.1, (M, 256, 768 // M)).astype(np.float32))
    block = rng.integers(0, 256, (1 << 20, M), dtype=np.uint8)
    return sizes, A, cent, pqc, block


def synthetic_pq_shard(n_codes: int, nlist: int, M: int = 96, device: int = 0, seed: int = 0, return_parts: bool = False,
                       doc_len: int = 0, skew: str = "", hot: float = 0.18):
    """A PQ index resident in HBM for TIMING the IVFPQ search (csrc/dph_pq.hip) at full size: random 8-bit codes, random
    codebooks and coarse centroids, list lengths with exponential weights (or a ``skew``: see ``_pq_list_sizes`` / ``synthetic_pq_parts``), a
    Householder reflection x permutation as the OPQ matrix -- what the scan costs does not depend on what the codes mean (parity: tests/test_pq.py on trained indexes).
    ``doc_len`` > 0: idx2id / f2o of documents of ``doc_len`` consecutive ids with every token kept (what SynthDocStore describes), so
    that MIPS.search runs end to end over the index; 0: a single stand-in document (search only).
    Returns (shard, A, centroids, list_sizes) (+ codebooks and the code block with ``return_parts``)."""
    import ctypes as C
    from . import _lib
    sizes, A, cent, pqc, block = synthetic_pq_parts(n_codes, nlist, M, seed, skew, hot)
    s = _lib.Shard.__new__(_lib.Shard)
    s._h = C.c_void_p()
    _lib._chk(_lib.lib.dph_index_create_pq(int(device), int(n_codes), int(nlist), int(M), C.byref(s._h)))
    s.device, s.id_base, s.n_rows = int(device), 0, int(n_codes)
    _lib._chk(_lib.lib.dph_index_set_pq(s._h, _lib._p(A), None, _lib._p(cent), _lib._p(pqc), 1))
    _lib._chk(_lib.lib.dph_index_set_pq_list_sizes(s._h, _lib._p(sizes)))
    # positions are list-major and contiguous: upload in chunks of the random block, whatever list they fall into
    for pos in range(0, n_codes, block.shape[0]):
        m = min(block.shape[0], n_codes - pos)
        c = np.ascontiguousarray(np.roll(block, pos // block.shape[0] % 97, axis=0)[:m])
        ids = np.arange(pos, pos + m, dtype=np.int64)
        _lib._chk(_lib.lib.dph_index_upload_pq_codes(s._h, pos, m, _lib._p(c), _lib._p(ids)))
    s.pq = {"nlist": int(nlist), "M": int(M), "nprobe": 256}
    if doc_len > 0:
        ids = np.arange(n_codes, dtype=np.int64)
        s.set_idx2id((ids // doc_len).astype(np.int32), (ids % doc_len).astype(np.int32))
        n_docs = (n_codes + doc_len - 1) // doc_len
        s.set_f2o(np.arange(n_docs, dtype=np.int32), np.arange(0, (n_docs + 1) * doc_len, doc_len, dtype=np.int64),
                  np.tile(np.arange(doc_len, dtype=np.int32), n_docs))
        del ids
    else:
        s.set_idx2id(np.zeros(n_codes, np.int32), np.zeros(n_codes, np.int32))
        s.set_f2o(np.zeros(1, np.int32), np.asarray([0, 1], np.int64), np.zeros(1, np.int32))
    s.finalize()
    if return_parts:
        return s, A, cent, sizes, pqc, block
    return s, A, cent, sizes
