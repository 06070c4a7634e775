"""Start/end-vector scoring of the reference's ``encoder.py`` on libdph's HIP kernels:

  ``phrase_logits``  -- ``Encoder.train_query`` (/root/reference/densephrases/encoder.py:383-386): the query's CLS vectors
                        against the ``[B, 2k, 768]`` start / end vectors ``MIPS.search(return_idxs=True)`` returned;
  ``dense_logits``   -- ``Encoder.forward`` (:206-208): every token's start / end vector against the query, and the
                        ``[bs, T, T]`` span-score table.

Torch tensors on the GPU in, torch tensors out, differentiable in every input like the torch lines they replace: the
query gradient is the HIP kernel (query-side fine-tuning back-propagates into the query encoder, train_query.py:208-275;
there the phrase vectors come from the index and need none), the gradient w.r.t. the vectors (``Encoder.forward`` is
training code: ``start`` / ``end`` come from the phrase encoder and ``dense_logits`` feeds the loss, encoder.py:206-208)
is the outer product grad[b,m] * q[b], the span table's gradient the two marginal sums.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _ptr(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())


def _check(q: torch.Tensor, vecs: torch.Tensor):
    if not (q.is_cuda and vecs.is_cuda and q.device == vecs.device):
        raise ValueError("score_vecs: GPU tensors on one device (libdph has no CPU path)")
    if q.dim() != 2 or vecs.dim() != 3 or q.shape[1] != _lib.DIM or vecs.shape[2] != _lib.DIM or q.shape[0] != vecs.shape[0]:
        raise ValueError(f"score_vecs: q [B,{_lib.DIM}] and vecs [B,M,{_lib.DIM}] expected, got {tuple(q.shape)} / {tuple(vecs.shape)}")


class _ScoreVecs(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, vecs):
        q32, v32 = q.detach().contiguous().float(), vecs.detach().contiguous().float()
        out = torch.empty(v32.shape[:2], dtype=torch.float32, device=q.device)
        st = torch.cuda.current_stream(q.device).cuda_stream
        _lib._chk(_lib.lib.dph_score_vecs_dev(q.device.index, _ptr(q32), _ptr(v32), v32.shape[0], v32.shape[1], _ptr(out),
                                              C.c_void_p(st)))
        ctx.save_for_backward(v32, q32)
        ctx.q_dtype, ctx.v_dtype = q.dtype, vecs.dtype
        return out

    @staticmethod
    def backward(ctx, grad):
        v32, q32 = ctx.saved_tensors
        g = grad.contiguous().float()
        gq = gv = None
        if ctx.needs_input_grad[0]:
            gq = torch.empty((v32.shape[0], _lib.DIM), dtype=torch.float32, device=g.device)
            st = torch.cuda.current_stream(g.device).cuda_stream
            _lib._chk(_lib.lib.dph_score_vecs_bwd_dev(g.device.index, _ptr(g), _ptr(v32), v32.shape[0], v32.shape[1], _ptr(gq),
                                                      C.c_void_p(st)))
            gq = gq.to(ctx.q_dtype)
        if ctx.needs_input_grad[1]:
            gv = (g.unsqueeze(-1) * q32.unsqueeze(1)).to(ctx.v_dtype)          # d<q, v_m>/dv_m = q
        return gq, gv


class _DenseLogits(torch.autograd.Function):
    """out[b,i,j] = s[b,i] + e[b,j] (encoder.py:208) on dph_dense_logits_dev; backward = the two marginal sums"""

    @staticmethod
    def forward(ctx, s, e):
        s32, e32 = s.detach().contiguous().float(), e.detach().contiguous().float()
        out = torch.empty((s32.shape[0], s32.shape[1], s32.shape[1]), dtype=torch.float32, device=s.device)
        st = torch.cuda.current_stream(s.device).cuda_stream
        _lib._chk(_lib.lib.dph_dense_logits_dev(s.device.index, _ptr(s32), _ptr(e32), s32.shape[0], s32.shape[1], _ptr(out),
                                                C.c_void_p(st)))
        ctx.dtypes = (s.dtype, e.dtype)
        return out

    @staticmethod
    def backward(ctx, grad):
        gs = grad.sum(2).to(ctx.dtypes[0]) if ctx.needs_input_grad[0] else None
        ge = grad.sum(1).to(ctx.dtypes[1]) if ctx.needs_input_grad[1] else None
        return gs, ge


def score_vecs(q: torch.Tensor, vecs: torch.Tensor) -> torch.Tensor:
    """out[b, m] = <q[b], vecs[b, m]>: q [B,768], vecs [B,M,768] -> [B,M] fp32."""
    _check(q, vecs)
    return _ScoreVecs.apply(q, vecs)


def phrase_logits(query_start: torch.Tensor, query_end: torch.Tensor, start_vecs: torch.Tensor, end_vecs: torch.Tensor):
    """encoder.py:383-386.  query_* [B,1,768] (or [B,768]), *_vecs [B,M,768] -> (start_logits, end_logits, logits) [B,M]."""
    qs = query_start.reshape(query_start.shape[0], -1)
    qe = query_end.reshape(query_end.shape[0], -1)
    s, e = score_vecs(qs, start_vecs), score_vecs(qe, end_vecs)
    return s, e, s + e


def dense_logits(start: torch.Tensor, end: torch.Tensor, query_start: torch.Tensor, query_end: torch.Tensor):
    """encoder.py:206-208.  start / end [bs,T,768], query_* [bs,1,768] -> (start_logits [bs,T], end_logits [bs,T],
    dense_logits [bs,T,T])."""
    qs = query_start.reshape(query_start.shape[0], -1)
    qe = query_end.reshape(query_end.shape[0], -1)
    s, e = score_vecs(qs, start), score_vecs(qe, end)
    return s, e, _DenseLogits.apply(s, e)
