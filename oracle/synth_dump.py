"""Synthetic phrase dumps in the reference's logical layout (test infrastructure).

Layout follows /root/reference/densephrases/utils/embed_utils.py:235-246 (one group per
document: context with ' [PAR] ' separators, title, int8 ``start`` rows of the *filtered* tokens,
``f2o_start`` filtered->original token index, ``word2char_start/end`` original token -> char
offsets) and build_phrase_index.py:192-276 for the row / idx2id order.
"""
from __future__ import annotations

from typing import List

import numpy as np

from .mips_oracle import DocMeta, float_to_int8

_VOCAB = ("alpha bravo charlie delta echo foxtrot golf hotel india juliet kilo lima mike november oscar "
          "papa quebec romeo sierra tango uniform victor whiskey xray yankee zulu The A An of in on").split()


def make_doc(rng: np.random.Generator, doc_idx: int, d: int = 768, n_par: int = 3, words_per_par=(8, 30),
             keep_prob: float = 0.75, sigma: float = 0.6) -> DocMeta:
    pars: List[List[str]] = []
    for _ in range(n_par):
        n = int(rng.integers(words_per_par[0], words_per_par[1] + 1))
        pars.append([_VOCAB[int(i)] for i in rng.integers(0, len(_VOCAB), n)])
    context = " [PAR] ".join(" ".join(p) + "." for p in pars)
    w2c_s, w2c_e = [], []
    pos = 0
    for pi, p in enumerate(pars):
        for wi, w in enumerate(p):
            w2c_s.append(pos)
            end = pos + len(w) + (1 if wi == len(p) - 1 else 0)    # trailing '.' belongs to the last word
            w2c_e.append(end)
            pos = end + 1 if wi < len(p) - 1 else end
        pos += len(" [PAR] ")
    n_tok = len(w2c_s)
    keep = rng.random(n_tok) < keep_prob
    if not keep.any():
        keep[0] = True
    f2o = np.nonzero(keep)[0].astype(np.int64)
    start = float_to_int8(rng.normal(0.0, sigma, size=(f2o.size, d)).astype(np.float32))
    return DocMeta(doc_idx=doc_idx, title=f"Doc {doc_idx}", context=context, f2o_start=f2o,
                   word2char_start=np.asarray(w2c_s, np.int32), word2char_end=np.asarray(w2c_e, np.int32),
                   start=start)


def make_dump(seed: int = 42, n_docs: int = 6, d: int = 768, doc_ids=None, **kw) -> List[DocMeta]:
    rng = np.random.default_rng(seed)
    if doc_ids is None:
        doc_ids = [int(x) for x in sorted(rng.choice(900, size=n_docs, replace=False))]
    return [make_doc(rng, di, d=d, **kw) for di in doc_ids]


def make_queries(rng: np.random.Generator, xb: np.ndarray, B: int, noise: float = 0.1) -> np.ndarray:
    """[B, 2d] queries = (a stored row, a row a few tokens later) de-quantised + N(0, noise^2):
    the truth is known."""
    N, d = xb.shape
    s = rng.integers(0, N, B)
    e = np.minimum(s + rng.integers(0, 4, B), N - 1)
    qs = xb[s].astype(np.float32) / 20.0 - 2.0 + rng.normal(0, noise, (B, d)).astype(np.float32)
    qe = xb[e].astype(np.float32) / 20.0 - 2.0 + rng.normal(0, noise, (B, d)).astype(np.float32)
    return np.concatenate([qs, qe], 1).astype(np.float32)
