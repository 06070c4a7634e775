"""N > 1 path on CPU: world_size-2 gloo run of the shard exchange (record packing, one all-gather, merge, following
the winners back into the gathered window results).  The local engine and the merge are oracle/numpy stand-ins here
(test infrastructure); on the GPU the same exchange code is driven by libdph (densephrases_amd.dist.ShardedSearcher)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from densephrases_amd.dist import RecordLayout, exchange_and_merge, partition_rows
from oracle import mips_oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _merge_cpu(va):
    D, I = va["D"].numpy(), va["I"].numpy()                       # [world, n, k]
    world, n, k = D.shape
    Dg = np.full((n, k), -O.FLT_MAX, np.float32)
    Ig = np.full((n, k), -1, np.int64)
    src = np.full((n, k), -1, np.int32)
    for r in range(n):
        d, i = D[:, r, :].reshape(-1), I[:, r, :].reshape(-1)
        ok = np.nonzero(i >= 0)[0]
        order = ok[np.lexsort((i[ok], -d[ok].astype(np.float64)))][:k]
        Dg[r, :order.size], Ig[r, :order.size], src[r, :order.size] = d[order], i[order], order
    return torch.from_numpy(Dg), torch.from_numpy(Ig), torch.from_numpy(src)


def _worker(rank, world, port, seed, n_rows, B, k, L, out_path, staged=False):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    coll = dist
    if staged:
        # the adapter the one-GPU multi-process runs use (collectives staged through host memory): on host tensors it must be
        # transparent
        from densephrases_amd.dist import HostStagedCollectives
        coll = HostStagedCollectives()
        assert coll.get_rank() == rank and coll.get_world_size() == world
        t = torch.tensor([float(rank + 1)])
        coll.all_reduce(t)
        assert float(t) == world * (world + 1) / 2
        t = torch.tensor([float(rank)])
        coll.all_reduce(t, op=coll.ReduceOp.MAX)
        assert float(t) == world - 1
        outs = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        coll.all_gather(outs, torch.tensor([rank, 10 * rank]))
        assert [o.tolist() for o in outs] == [[r, 10 * r] for r in range(world)]
    rng = np.random.default_rng(seed)
    xb = O.float_to_int8(rng.normal(0, 0.6, (n_rows, 768)).astype(np.float32))
    xb[n_rows // 2 + 3] = xb[5]                                  # a cross-shard exact tie
    q = rng.normal(0, 0.5, (2 * B, 768)).astype(np.float32)
    doc_len = 50
    lo, hi = partition_rows(n_rows, world, align=doc_len)[rank]
    # local engine (oracle): top-k over the slice with global ids, then a fake window result keyed by id
    D, I, _ = O.flat_ip_search(q, xb[lo:hi], k, id_base=lo)
    layout = RecordLayout(2 * B, k)
    rec = torch.zeros(layout.nbytes, dtype=torch.uint8)
    rec_all = torch.zeros((world, layout.nbytes), dtype=torch.uint8)
    v = layout.views(rec)
    v["D"].copy_(torch.from_numpy(D))
    v["I"].copy_(torch.from_numpy(I))
    v["best"].copy_(torch.from_numpy(I.astype(np.float64) * 0.5 + 1.0))       # payload derived from the id
    v["pred"].copy_(torch.from_numpy((I % 1000).astype(np.int32)))
    v["status"].fill_(rank == 1 and 1 or 0)
    Dg, Ig, best, pred, status = exchange_and_merge(layout, rec, rec_all, coll, world, _merge_cpu)
    if rank == 0:
        np.savez(out_path, D=Dg.numpy(), I=Ig.numpy(), best=best.numpy(), pred=pred.numpy(), status=status.numpy(),
                 q=q, xb=xb)
    coll.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_rows,k,staged", [(1000, 10, False), (60, 40, False), (1000, 10, True)])
def test_two_rank_exchange_matches_single_index(tmp_path, n_rows, k, staged):
    world, B, L = 2, 4, 10
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(world, _free_port(), 3, n_rows, B, k, L, out, staged), nprocs=world, join=True)
    z = np.load(out)
    Dr, Ir, D64 = O.flat_ip_search(z["q"], z["xb"], k)
    ok, msg = O.topk_equivalent(z["D"], z["I"], D64, Ir)
    assert ok, msg
    valid = z["I"] >= 0
    np.testing.assert_array_equal(z["best"][valid], z["I"][valid] * 0.5 + 1.0)      # payload followed its id
    np.testing.assert_array_equal(z["pred"][valid], z["I"][valid] % 1000)
    assert (z["pred"][~valid] == -1).all()
    assert (z["status"] == 1).all()                                                 # max over ranks


def test_partition_rows():
    parts = partition_rows(170_000_000, 8)
    assert parts[0][0] == 0 and parts[-1][1] == 170_000_000
    assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
    assert all(lo % 800 == 0 for lo, _ in parts)
    sizes = [hi - lo for lo, hi in parts]
    assert max(sizes) - min(sizes) <= 1600
    starts = np.array([0, 10, 25, 60, 61, 90])
    p = partition_rows(100, 3, doc_starts=starts)
    assert [lo for lo, _ in p] == [0, 60, 90] and p[-1][1] == 100
    assert partition_rows(5, 1) == [(0, 5)]


def test_record_layout_views_alias_the_buffer():
    lay = RecordLayout(6, 3)
    buf = torch.zeros(lay.nbytes, dtype=torch.uint8)
    v = lay.views(buf)
    v["I"][2, 1] = 123456789012
    v["D"][5, 2] = 1.5
    v["status"][4] = 7
    v2 = lay.views(buf.clone().unsqueeze(0).repeat(2, 1))
    assert v2["I"].shape == (2, 6, 3) and int(v2["I"][1, 2, 1]) == 123456789012
    assert float(v2["D"][0, 5, 2]) == 1.5 and int(v2["status"][1, 4]) == 7
    assert all(off % 8 == 0 for off, _ in lay.fields.values())


def test_a_rank_that_cannot_set_the_agreed_aux_layout_makes_every_rank_raise():
    """dist.sync_aux_layout: the ranks all-gather their layouts, agree, set -- and all-gather whether the set worked.  A rank
    whose shard refuses the agreed layout (its row norms do not fit the clamp) used to raise alone while the others went on into the
    next collective and hung; now EVERY rank raises.  Ranks = threads with stub shards, collectives through a barrier."""
    import threading
    import torch
    from densephrases_amd.dist import sync_aux_layout

    world = 3
    slots, bar = [None] * world, threading.Barrier(world)

    class Dist:
        def __init__(self, rank):
            self.rank = rank

        def all_gather_into_tensor(self, out, inp):
            slots[self.rank] = inp.clone()
            bar.wait()
            o = out.view(world, -1)
            for r in range(world):
                o[r].copy_(slots[r].reshape(-1))
            bar.wait()

    class Shard:
        pq = None

        def __init__(self, rank, refuse):
            lay = np.full(28, -1, np.int32)
            lay[:4] = (16, 4, 2, 64) if rank == 0 else (0, 0, 0, 40 + rank)
            if rank == 0:
                lay[4:6] = (77, 138)
            self.lay, self.refuse, self.set_to = lay, refuse, None

        def aux_layout(self):
            return self.lay.copy()

        def set_aux_layout(self, lay):
            if self.refuse:
                raise RuntimeError("q2max too large for this shard's row norms")
            self.set_to = np.asarray(lay).copy()

    for refuse_rank in (None, 2):
        shards = [Shard(r, r == refuse_rank) for r in range(world)]
        errs = [None] * world

        def run(r):
            try:
                sync_aux_layout(shards[r], Dist(r), world, torch.device("cpu"))
            except RuntimeError as e:
                errs[r] = str(e)

        ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=60)
        if refuse_rank is None:
            assert errs == [None] * world
            assert all(s._aux_synced == world for s in shards)
            # the widest stride, its replica table, the smallest clamp -- on the ranks that had chosen otherwise
            assert shards[0].set_to is not None and shards[0].set_to[3] == 41 and shards[1].set_to[0] == 16 and list(shards[2].set_to[4:6]) == [77, 138]
        else:
            assert all(e is not None and "rank(s) [2]" in e for e in errs), errs
            assert not any(getattr(s, "_aux_synced", 0) == world for s in shards)
