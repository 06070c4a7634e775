"""Worker of tests/test_dist_nccl.py::test_one_rank_nccl_group_carries_the_real_exchange_buffers: ONE process, ONE GPU, a 1-rank
`nccl` (= RCCL) process group, and the sharded path forced through it -- the aux-layout all-gather (int32), the sample all-gather
(int32 [2B, 16]), the packed record all-gather (uint8), the return_idxs SUM all-reduce (fp32) -- against the plain single-rank
path.  Exercises what gloo cannot: RCCL communicator init, dtype support, ordering against the search's stream.  Exits non-zero on
any mismatch."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def same(a, b):
    ok = len(a) == len(b)
    for g, w in zip(a, b):
        ok &= len(g) == len(w)
        for x, y in zip(g, w):
            ok &= all(x[k] == y[k] for k in ("doc_idx", "start_idx", "end_idx", "answer", "score"))
            if y.get("start_vec") is not None:
                ok &= bool(np.array_equal(x["start_vec"], y["start_vec"]) and np.array_equal(x["end_vec"], y["end_vec"]))
    return bool(ok)


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl"
    from densephrases_amd import DocMeta, DocStore, MIPS, Shard
    from densephrases_amd.dist import ShardedSearcher
    from oracle.synth_dump import make_dump, make_queries
    docs = make_dump(seed=11, n_docs=300, d=768, n_par=4, words_per_par=(20, 40))
    conv = lambda: DocStore([DocMeta(m.doc_idx, m.title, m.context, m.f2o_start, m.word2char_start, m.word2char_end,  # noqa: E731
                                     m.start) for m in docs])
    forced = MIPS(None, "in-memory", None, device=0, _store=conv(), rank=0, world=1, dist=dist, force_collectives=True)
    plain = MIPS(None, "in-memory", None, device=0, _store=conv(), rank=0, world=1)
    q = make_queries(np.random.default_rng(4), conv().rows, 12)
    texts = [f"q{i}" for i in range(12)]
    kw = dict(top_k=10, aggregate=True, agg_strat="opt1")
    ok = same(forced.search(q, q_texts=texts, **kw), plain.search(q, q_texts=texts, **kw))
    ok &= same(forced.search(q[:4], q_texts=texts[:4], top_k=5, return_idxs=True),
               plain.search(q[:4], q_texts=texts[:4], top_k=5, return_idxs=True))
    ok &= same(list(forced.search_stream([q, q[:6]], q_texts=[texts, texts[:6]], **kw))[1],
               plain.search(q[:6], q_texts=texts[:6], **kw))
    # the device-resident step of bench.py over a shard with aux rows (its layout crosses RCCL too), union bound on
    n, B, k, L = 300_000, 16, 10, 10
    s = Shard(n, device=0)
    s.fill_synthetic(seed=5, kind=4)
    s.set_idx2id((np.arange(n) // 100).astype(np.int32), (np.arange(n) % 100).astype(np.int32))
    s.set_f2o(np.arange(n // 100, dtype=np.int32), np.arange(0, n + 1, 100, dtype=np.int64), np.tile(np.arange(100, dtype=np.int32), n // 100))
    s.finalize()
    qd = torch.from_numpy(np.random.default_rng(1).normal(0, 0.5, (B, 1536)).astype(np.float32)).to(dev)
    a = ShardedSearcher(s, B, k, L, rank=0, world=1, dist=dist, device=dev, force_collectives=True)
    assert a.coll and a.union_bounds and getattr(s, "_aux_synced", 0) == 1
    got = {kk: v.clone() for kk, v in a.step(qd).items()}
    b = ShardedSearcher(s, B, k, L, device=dev)
    want = b.step(qd)
    torch.cuda.synchronize()
    for kk in ("D", "I", "best", "pred", "status"):
        ok &= bool(torch.equal(got[kk], want[kk]))
    ok &= int(want["status"].max()) == 0
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
