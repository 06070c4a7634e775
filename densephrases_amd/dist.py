"""Range-sharded multi-GPU search: one process per GPU, the phrase dump cut into contiguous id ranges at document
boundaries, per-GPU local top-k + local window re-score, then ONE all-gather (RCCL over xGMI through
``torch.distributed``) of every rank's [2B, k] record and an on-device (score desc, id asc) merge.

The reference has no counterpart (its inference path is single-process; only the coarse quantizer is replicated,
/root/reference/densephrases/index.py:52-56) -- this is SURVEY.md section 8(e).  The record is tiny
(2B*k*(4+8+8+4) B = 30 KB at B=64, k=10), so the exchange is latency-bound: a single collective per batch and no
ring all-reduce anywhere.

The local engine and the merge are injected, so the exchange logic is exercised on CPU with ``gloo`` in the tests
(tests/test_dist_gloo.py) with an oracle-backed engine; the product wiring (``ShardedSearcher``) always uses libdph.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import numpy as np


def partition_rows(n_total: int, world: int, align: int = 800,
                   doc_starts: Optional[np.ndarray] = None) -> List[Tuple[int, int]]:
    """Contiguous [lo, hi) row ranges, balanced by row count.  Cuts are moved to document boundaries when
    ``doc_starts`` (sorted first-row index of every document) is given -- windows never leave a document
    (index.py:305-321), so doc-aligned shards need no halo -- otherwise to multiples of ``align``."""
    cuts = [0]
    for r in range(1, world):
        c = (n_total * r) // world
        if doc_starts is not None and len(doc_starts):
            j = int(np.searchsorted(doc_starts, c, side="left"))
            c = int(doc_starts[j]) if j < len(doc_starts) else n_total
        else:
            c = (c // align) * align
        cuts.append(max(cuts[-1], min(c, n_total)))
    cuts.append(n_total)
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


def agree_aux_layout(layouts: np.ndarray) -> np.ndarray:
    """The aux layout every rank of a sharded job uses, from the layouts [world, 28] the ranks' shards chose for themselves
    (include/dph.h dph_index_get_aux_layout): the digits a query row is cut into depend on the replica table and the low-digit
    clamp, and the ranks exchange INTEGER scores, so those must be the same everywhere -- the widest stride, the replica table of the
    lowest rank that has one, the smallest clamp."""
    layouts = np.asarray(layouts, dtype=np.int32).reshape(-1, 28)
    out = np.zeros(28, dtype=np.int32)
    out[4:] = -1
    stride = int(layouts[:, 0].max())
    out[0] = stride
    out[3] = int(layouts[:, 3].min())
    if stride >= 16:
        src = layouts[np.nonzero(layouts[:, 0] == stride)[0][0]]
        out[1], out[2], out[4:] = src[1], src[2], src[4:]
    elif stride == 4:
        out[1] = 4
    return out


def sync_aux_layout(shard, dist, world: int, device, force: bool = False) -> None:
    """Collective: make every rank's shard cut its queries into the same digits (once per shard and job)."""
    import torch
    if (world <= 1 and not force) or dist is None or getattr(shard, "pq", None) or getattr(shard, "_aux_synced", 0) == world:
        return
    mine = shard.aux_layout()
    t = torch.from_numpy(mine).to(device)
    allr = torch.empty((world, mine.size), dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(allr.view(-1), t)
    agreed = agree_aux_layout(allr.cpu().numpy())
    err = None
    if not np.array_equal(agreed, mine):
        try:
            shard.set_aux_layout(agreed)
        except Exception as e:                      # noqa: BLE001  (raised below, on EVERY rank)
            err = e
    # every rank learns whether every rank could set the layout: one that cannot must not leave the others waiting in the next collective
    ok = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=device)
    oks = torch.empty(world, dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(oks, ok)
    bad = torch.nonzero(oks).flatten().cpu().tolist()
    if bad:
        raise RuntimeError(f"sync_aux_layout: rank(s) {bad} could not set the agreed aux layout {agreed[:4].tolist()}" + (f": {err}" if err is not None else ""))
    shard._aux_synced = world


class RecordLayout:
    """Byte layout of one rank's exchange record: D f32 [n,k] | I i64 [n,k] | best f64 [n,k] | pred i32 [n,k] |
    status i32 [n] | bound f64 [n] (upper bound of the score of any row the rank did not return; -1e300 = none);
    every field starts on an 8-byte boundary."""

    def __init__(self, n_rows: int, k: int):
        self.n, self.k = n_rows, k
        off = 0
        self.fields = {}
        for name, itemsize, count in (("D", 4, n_rows * k), ("I", 8, n_rows * k), ("best", 8, n_rows * k),
                                      ("pred", 4, n_rows * k), ("status", 4, n_rows), ("bound", 8, n_rows)):
            self.fields[name] = (off, itemsize * count)
            off += (itemsize * count + 7) // 8 * 8
        self.nbytes = off

    def views(self, buf):
        """torch views of a uint8 buffer [nbytes] (or [world, nbytes] -> leading world dim)."""
        import torch
        dt = {"D": torch.float32, "I": torch.int64, "best": torch.float64, "pred": torch.int32, "status": torch.int32,
              "bound": torch.float64}
        out = {}
        for name, (off, nb) in self.fields.items():
            v = buf[..., off:off + nb].view(dt[name])
            shape = (self.n, self.k) if name not in ("status", "bound") else (self.n,)
            out[name] = v.reshape(*buf.shape[:-1], *shape)
        return out


def exchange_and_merge(layout: RecordLayout, rec, rec_all, dist, world: int, merge_fn: Callable, collective: Optional[bool] = None):
    """all-gather the per-rank records and merge.  ``merge_fn(rec_all_views)`` returns either ``(D, I, src)`` with
    src = part*k+col (the winners are then followed into the gathered window results here, in torch -- the form the
    CPU tests inject) or the finished ``(D, I, best, pred, status)`` (libdph's fused dph_merge_records_dev)."""
    import torch
    if (world > 1) if collective is None else collective:
        dist.all_gather_into_tensor(rec_all.view(-1), rec)
    else:
        rec_all.view(-1).copy_(rec)
    va = layout.views(rec_all)
    merged = merge_fn(va)
    if len(merged) == 5:
        return merged
    D, I, src = merged
    # follow the winners back into the gathered window results
    n, k = layout.n, layout.k
    srcl = src.to(torch.int64).clamp_min(0)
    part, col = srcl // k, srcl % k
    rows = torch.arange(n, device=src.device).unsqueeze(1).expand(n, k)
    best = va["best"][part, rows, col]
    pred = va["pred"][part, rows, col]
    best = torch.where(src >= 0, best, torch.full_like(best, -1e9))
    pred = torch.where(src >= 0, pred, torch.full_like(pred, -1))
    # (dph_merge_kernel's rule: 3 = the ranks flagged a non-finite query row -- they all see the same row --, else 1 unless every part is 0)
    nf = (va["status"] == 3).any(dim=0)
    status = torch.where(nf, torch.full_like(nf, 3, dtype=torch.int32), ((va["status"] != 0) & (va["status"] != 3)).any(dim=0).to(torch.int32))
    return D, I, best, pred, status


class HostStagedCollectives:
    """The slice of ``torch.distributed`` this package uses (all_gather_into_tensor, all_reduce, barrier, rank / world),
    for DEVICE tensors over a backend that only takes host tensors (gloo): every collective is staged through host
    memory.  Lets several processes that share ONE GPU run the sharded path across real process boundaries
    (tests/test_dist_one_gpu.py, ``DPH_BENCH_ONE_GPU=1 python bench.py --gpus N``); production runs use RCCL directly."""

    def __init__(self):
        import torch.distributed as dist
        self._d = dist
        self.ReduceOp = dist.ReduceOp

    def all_gather_into_tensor(self, out, inp):
        import torch
        o, i = torch.empty(out.shape, dtype=out.dtype), inp.detach().cpu()
        self._d.all_gather_into_tensor(o, i)
        out.copy_(o)

    def all_gather(self, outs, inp):
        import torch
        host = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
        self._d.all_gather(host, inp.detach().cpu())
        for o, h in zip(outs, host):
            o.copy_(h)

    def all_reduce(self, t, op=None):
        c = t.detach().cpu()
        self._d.all_reduce(c, op=op if op is not None else self._d.ReduceOp.SUM)
        t.copy_(c)

    def barrier(self):
        self._d.barrier()

    def get_rank(self):
        return self._d.get_rank()

    def get_world_size(self):
        return self._d.get_world_size()

    def destroy_process_group(self):
        self._d.destroy_process_group()


class ShardedSearcher:
    """The timed hot path of bench.py / the device-resident serving loop: search + window re-score of one batch on the
    local shard, exchange + merge across ranks.  Everything stays on the GPU; buffers are allocated once."""

    def __init__(self, shard, B: int, k: int, L: int, rank: int = 0, world: int = 1, dist=None, device=None,
                 union_bounds: Optional[bool] = None, force_collectives: bool = False):
        """``union_bounds`` (default: on when world > 1): two-phase search -- every rank shares the 16 best scores of
        its pre-pass sample per query row (a second small all-gather), all ranks scan under the bound of the UNION of
        the samples (8x fewer rare-path entries per shard at 8 ranks), and the exactness certificate is taken after
        the merge (include/dph.h: dph_search_sample_dev / dph_union_bounds_dev / dph_search_bounded_dev).
        ``force_collectives``: a job of ONE rank takes the multi-rank path all the same -- union bound, both all-gathers through
        ``dist``, the merge of one record -- so that the transport (RCCL) is exercised on a 1-GPU box."""
        import torch
        self.shard, self.B, self.k, self.L = shard, B, k, L
        self.rank, self.world, self.dist = rank, world, dist
        self.coll = world > 1 or (bool(force_collectives) and dist is not None)
        self.dev = device if device is not None else torch.device("cuda", shard.device)
        n = 2 * B
        sync_aux_layout(shard, dist, world, self.dev, force=self.coll)
        self.layout = RecordLayout(n, k)
        self.x = torch.empty((n, 768), dtype=torch.float32, device=self.dev)
        self.union_bounds = self.coll if union_bounds is None else bool(union_bounds)
        if getattr(shard, "pq", None):
            # a PQ shard has no sampled bounds to share: every rank probes the same lists, scores its own share of their codes, and
            # the ranks' top-k merge to the single-GPU answer
            self.union_bounds = False
        if self.union_bounds and world > 1:
            self.set_sample_world(world)
        self.top = torch.empty((n, 16), dtype=torch.int32, device=self.dev)
        self.top_all = torch.empty((world, n, 16), dtype=torch.int32, device=self.dev)
        self.tau = torch.empty((n,), dtype=torch.int32, device=self.dev)
        self.rec = torch.zeros(self.layout.nbytes, dtype=torch.uint8, device=self.dev)
        self.rec_all = torch.zeros((world, self.layout.nbytes), dtype=torch.uint8, device=self.dev)
        self.v = self.layout.views(self.rec)
        self.arg = torch.empty((n, k), dtype=torch.int32, device=self.dev)
        # the merged result is laid out like a record too, so a caller can take it to the host in one copy (`result_record`)
        self.rec_out = torch.zeros(self.layout.nbytes, dtype=torch.uint8, device=self.dev)
        vo = self.layout.views(self.rec_out)
        self.Dg, self.Ig, self.bestg, self.predg, self.statusg = vo["D"], vo["I"], vo["best"], vo["pred"], vo["status"]
        # measurement (bench.py --gpus N): event pairs around the two exchanges of a step, read with ``collective_ms``
        self.time_collectives = False
        self._coll_events = {"sample_all_gather": [], "record_all_gather_and_merge": []}

    def _timed(self, name, fn):
        """run ``fn`` between two events on the current stream when ``time_collectives`` is on (what the stream WAITS for an exchange:
        the collective itself plus the arrival of the slowest rank)"""
        import torch
        if not self.time_collectives:
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        self._coll_events[name].append((a, b))
        return out

    def collective_ms(self):
        """{exchange: (summed ms, count)} since the last call; synchronises the recorded events"""
        out = {}
        for name, evs in self._coll_events.items():
            tot = 0.0
            for a, b in evs:
                b.synchronize()
                tot += a.elapsed_time(b)
            out[name] = (tot, len(evs))
            evs.clear()
        return out

    def _merge(self, va):
        from . import _lib
        import torch
        st = torch.cuda.current_stream(self.dev).cuda_stream
        _lib.merge_records_dev(self.shard.device, va["D"].data_ptr(), va["I"].data_ptr(), va["best"].data_ptr(),
                               va["pred"].data_ptr(), va["status"].data_ptr(), self.world, 2 * self.B, self.k,
                               self.Dg.data_ptr(), self.Ig.data_ptr(), self.bestg.data_ptr(), self.predg.data_ptr(),
                               self.statusg.data_ptr(), stream=st, part_stride_bytes=self.layout.nbytes,
                               bound_ptr=va["bound"].data_ptr())
        return self.Dg, self.Ig, self.bestg, self.predg, self.statusg

    def set_sample_world(self, world: int):
        """Under a union bound the sample that matters is the UNION of the ranks' samples: the bound is the 16-th best
        of all of them, and ~16 * stride rows of the WHOLE dump beat it -- 16 * stride / world per rank.  A rank
        therefore samples `world` times more sparsely than a lone shard would (stride 32 * world) for the same number
        of emitted pairs, which takes the finest -- most expensive -- ladder level off every rank's step."""
        self.shard.set_tuning("fine_stride", int(min(32 * max(world, 1), 1024)))

    @property
    def result_record(self):
        """the packed record holding what the last ``step`` returned: this rank's own (world 1) or the merged one"""
        return self.rec if not self.coll else self.rec_out

    def load_query(self, q):
        """q: [B, 1536] fp32 on the device (start || end halves, index.py:196) -> the stacked [2B, 768] rows."""
        B = self.B
        self.x[:B].copy_(q[:, :768])
        self.x[B:].copy_(q[:, 768:])

    def sample(self):
        """Phase 1 of the two-phase search: this shard's 16 best pre-pass sample scores per row -> self.top."""
        import torch
        st = torch.cuda.current_stream(self.dev).cuda_stream
        self.shard.search_sample_dev(self.x.data_ptr(), 2 * self.B, self.top.data_ptr(), st)

    def union_bound(self, top_all, n_parts):
        """Per-row bound over the union of ``n_parts`` shards' samples ([n_parts, 2B, 16] int32) -> self.tau."""
        from . import _lib
        import torch
        st = torch.cuda.current_stream(self.dev).cuda_stream
        _lib.union_bounds_dev(self.shard.device, top_all.data_ptr(), n_parts, 2 * self.B, self.tau.data_ptr(), st)

    def search_and_rescore(self):
        """Local top-k (under self.tau when union_bounds) + both window passes into this rank's record."""
        import torch
        B, k, L, v, s = self.B, self.k, self.L, self.v, self.shard
        st = torch.cuda.current_stream(self.dev).cuda_stream
        if self.union_bounds:
            s.search_bounded_dev(self.x.data_ptr(), 2 * B, k, self.tau.data_ptr(), v["D"].data_ptr(), v["I"].data_ptr(),
                                 v["status"].data_ptr(), v["bound"].data_ptr(), st)
        else:
            s.search_dev(self.x.data_ptr(), 2 * B, k, v["D"].data_ptr(), v["I"].data_ptr(), v["status"].data_ptr(), st)
            if self.coll:
                v["bound"].fill_(-1.0e300)
        # find end for start candidates (rows [0,B)): END half of the query; find start for end candidates: START half
        self._rescore()

    def step(self, q):
        """q: [B, 1536] fp32 on the device.  Returns device tensors (merged over the ranks when world > 1)."""
        v = self.v
        self.load_query(q)
        if self.union_bounds:
            self.sample()
            if self.coll:
                self._timed("sample_all_gather", lambda: self.dist.all_gather_into_tensor(self.top_all.view(-1), self.top.view(-1)))
                self.union_bound(self.top_all, self.world)
            else:
                self.union_bound(self.top, 1)
        self.search_and_rescore()
        if not self.coll:
            return {"D": v["D"], "I": v["I"], "best": v["best"], "pred": v["pred"], "status": v["status"]}
        D, I, best, pred, status = self._timed("record_all_gather_and_merge", lambda: exchange_and_merge(
            self.layout, self.rec, self.rec_all, self.dist, self.world, self._merge, collective=True))
        return {"D": D, "I": I, "best": best, "pred": pred, "status": status}

    def _rescore(self):
        B, k, L, v, s = self.B, self.k, self.L, self.v, self.shard
        import torch
        st = torch.cuda.current_stream(self.dev).cuda_stream
        s.rescore_dev(0, self.x[B:].data_ptr(), B, k, L, v["I"][:B].data_ptr(), 0, 0, v["D"][:B].data_ptr(),
                      v["pred"][:B].data_ptr(), v["best"][:B].data_ptr(), self.arg[:B].data_ptr(), 0, st)
        s.rescore_dev(1, self.x[:B].data_ptr(), B, k, L, v["I"][B:].data_ptr(), 0, 0, v["D"][B:].data_ptr(),
                      v["pred"][B:].data_ptr(), v["best"][B:].data_ptr(), self.arg[B:].data_ptr(), 0, st)

    def step_exact(self, q):
        """``step`` plus the guarantee that no uncertified row leaves: libdph already retries on the device (own-bound
        re-scan, then the fp64 scan for up to 32 rows per call), so a row can only still be flagged when (a) more rows
        than that needed the fp64 scan, or (b) world > 1 and the certificate taken AFTER the merge failed (the merged
        k-th score does not beat the bound of a shard that answered under the union bound).  Those rows -- the same set
        on every rank, the merged status is identical everywhere -- are re-searched by every rank WITHOUT a union
        bound through the host entry point (exact local top-k, raises DphError if even that cannot be certified), the
        windows are recomputed and the records exchanged again.  Costs one device->host read of the status vector."""
        import torch
        out = self.step(q)
        bad = torch.nonzero(out["status"] == 1).flatten().cpu().numpy()      # (3 = a non-finite query row: final, ids -1)
        if bad.size == 0:
            return out
        v = self.v
        x = self.x.cpu().numpy()
        D, I = self.shard.search(x[bad], self.k)                 # exact over the local shard, certified or raises
        self.v["D"][bad] = torch.from_numpy(D).to(self.dev)
        self.v["I"][bad] = torch.from_numpy(I).to(self.dev)
        self.v["status"][bad] = 0
        if self.coll or self.union_bounds:
            self.v["bound"][bad] = -1.0e300
        self._rescore()
        if not self.coll:
            return {"D": v["D"], "I": v["I"], "best": v["best"], "pred": v["pred"], "status": v["status"]}
        D, I, best, pred, status = exchange_and_merge(self.layout, self.rec, self.rec_all, self.dist, self.world,
                                                      self._merge, collective=True)
        return {"D": D, "I": I, "best": best, "pred": pred, "status": status}


class PipelinedSearcher:
    """Two batches in flight on one flat shard (include/dph.h "two batches in flight"): while the full scan of batch t streams the
    dump on the main stream (all CUs but ``side_cus``), the latency-bound front of batch t+1 -- quantise, the sampled levels with
    their refine / threshold launches -- runs on a side stream that owns the remaining CUs; the tail of a batch (refine, select,
    retry chain, both window passes) follows its scan on the main stream.  Each batch in flight has a handle of its own
    (``Shard.twin()``: same rows, own scratch) and buffers of its own; the launches of a batch are exactly those of
    ``ShardedSearcher.step`` (world 1), so the results are identical -- one batch later: ``step(q)`` returns the result of the
    PREVIOUS batch (None for the first), ``flush()`` the last one.  Results are valid for the caller's current stream (it is made to
    wait for them; see ``_caller_stream`` for the default stream) until the step after next re-uses their lane.  ``q`` must not be
    overwritten on another stream before the step has copied it (a caller on its own stream is made to wait for that copy)."""

    def __init__(self, shard, B: int, k: int, L: int, side_cus: int = 8, device=None):
        import torch
        from . import _lib
        self.shard, self.B, self.k, self.L = shard, B, k, L
        self.dev = device if device is not None else torch.device("cuda", shard.device)
        cus = int(torch.cuda.get_device_properties(self.dev).multi_processor_count)
        if not 0 < side_cus < cus:
            raise ValueError("PipelinedSearcher: side_cus must leave CUs to the scan")
        self.side_cus, self.main_cus = int(side_cus), cus - int(side_cus)
        # (process-lifetime streams, shared by every searcher with the same split: torch's allocator may remember a stream a tensor was
        # used on, so a stream handed to torch is never destroyed)
        self.side = torch.cuda.ExternalStream(_lib.cu_range_stream(shard.device, 0, self.side_cus), device=self.dev)
        self.main = torch.cuda.ExternalStream(_lib.cu_range_stream(shard.device, self.side_cus, self.main_cus), device=self.dev)
        self.lanes = []
        for _ in range(2):
            t = shard.twin()
            t.set_tuning("scan_grid", self.main_cus)          # one persistent scan workgroup per CU of the main stream
            t.set_tuning("side_grid", self.side_cus)          # the sampled levels: one per CU of the side stream
            lane = ShardedSearcher(t, B, k, L, device=self.dev)
            lane.prep_done, lane.fin_done, lane.busy = torch.cuda.Event(), torch.cuda.Event(), False
            self.lanes.append(lane)
        self.t, self.pending = 0, None

    def _caller_stream(self):
        """The caller's current stream, or None when that is the device's default (NULL) stream: the CU-range streams are ordinary
        blocking streams, so work on the NULL stream is ordered against everything enqueued on them before and after by HIP itself
        -- and every command put on it, an event record or wait included, is a barrier across BOTH of them: one of those per step
        and nothing overlaps (measured: 42 ms per batch instead of 20).  A caller on the default stream therefore gets no event
        plumbing at all; its next NULL-stream operation (a copy to the host, say) waits for the results anyway."""
        import torch
        cur = torch.cuda.current_stream(self.dev)
        return None if cur == torch.cuda.default_stream(self.dev) else cur

    def _out(self, lane):
        if lane is None:
            return None
        cur = self._caller_stream()
        if cur is not None:
            cur.wait_event(lane.fin_done)
        v = lane.v
        return {"D": v["D"], "I": v["I"], "best": v["best"], "pred": v["pred"], "status": v["status"]}

    def step(self, q):
        """q: [B, 1536] fp32 on the device (ready on the caller's current stream).  Returns the previous batch's result."""
        import torch
        lane = self.lanes[self.t & 1]
        B, k = self.B, self.k
        cur = self._caller_stream()
        if cur is not None:
            ready = torch.cuda.Event()
            ready.record(cur)
        with torch.cuda.stream(self.side):
            if cur is not None:
                self.side.wait_event(ready)
            if lane.busy:
                self.side.wait_event(lane.fin_done)          # the batch before last has left this lane's scratch and buffers
            lane.load_query(q)
            loaded = torch.cuda.Event()
            loaded.record(self.side)
            lane.shard.search_prepare_dev(lane.x.data_ptr(), 2 * B, k, self.side.cuda_stream)
            lane.prep_done.record(self.side)
        with torch.cuda.stream(self.main):
            self.main.wait_event(lane.prep_done)
            v = lane.v
            lane.shard.search_finish_dev(lane.x.data_ptr(), 2 * B, k, v["D"].data_ptr(), v["I"].data_ptr(), v["status"].data_ptr(),
                                         self.main.cuda_stream)
            lane._rescore()                                    # (takes torch's current stream: the main stream here)
            lane.fin_done.record(self.main)
        if cur is not None:
            cur.wait_event(loaded)                         # q may be re-used by its producer once the lane holds a copy
        lane.busy = True
        prev, self.pending = self.pending, lane
        self.t += 1
        return self._out(prev)

    def flush(self):
        """The result of the last batch fed (None when there is none)."""
        prev, self.pending = self.pending, None
        return self._out(prev)

    def close(self):
        import torch
        torch.cuda.synchronize(self.dev)
        for lane in self.lanes:
            lane.shard.close()
        self.lanes = []
