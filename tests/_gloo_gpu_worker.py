"""Worker of tests/test_dist_one_gpu.py: WORLD processes that all use cuda:0 and exchange over gloo.  Every process runs
the PRODUCT path on its own range shard -- dph_search_sample_dev, dph_union_bounds_dev, dph_search_bounded_dev, the window
kernel, dph_merge_records_dev, step_exact -- and the records really cross a process boundary (the thread-rank tests of
test_gpu_search.py share one address space, test_dist_gloo.py has no GPU).  The collectives are the ones the RCCL path
issues (all_gather_into_tensor of the sample scores and of the packed record, the SUM all-reduce of the return_idxs
vectors); a thin adapter stages the device tensors through host memory because gloo does not take them.
Exits non-zero on any mismatch with a single-shard MIPS built by rank 0."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    torch.cuda.set_device(0)
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from densephrases_amd import DocMeta, DocStore, MIPS
    from densephrases_amd.dist import HostStagedCollectives
    from oracle.synth_dump import make_dump, make_queries
    docs = make_dump(seed=11, n_docs=300, d=768, n_par=4, words_per_par=(20, 40))
    conv = lambda: DocStore([DocMeta(m.doc_idx, m.title, m.context, m.f2o_start, m.word2char_start, m.word2char_end,  # noqa: E731
                                     m.start) for m in docs])
    # DPH_GLOO_WORKER_IVF=1: the shards are stored LIST-MAJOR (configs[3]: IVF with exact in-list inner product, specified on 4 GPUs) --
    # every rank builds the lists of its own rows over the SAME centroids and probes the same lists, so the merged answer is the
    # single-rank IVF answer
    ivf = None
    if os.environ.get("DPH_GLOO_WORKER_IVF") == "1":
        ivf = {"nlist": 32, "nprobe": 8, "centroids": np.random.default_rng(77).normal(0, 0.5, (32, 768)).astype(np.float32)}
    m = MIPS(None, "in-memory", None, device=0, _store=conv(), rank=rank, world=world, dist=HostStagedCollectives(), ivf=ivf)
    assert m.world == world and m.row_hi - m.row_lo < m.index.ntotal
    rows = conv().rows
    q = make_queries(np.random.default_rng(4), rows, 12)
    texts = [f"q{i}" for i in range(12)]
    got = m.search(q, q_texts=texts, top_k=10, aggregate=True, agg_strat="opt1")
    got_v = m.search(q[:4], q_texts=texts[:4], top_k=5, return_idxs=True)
    got_s = list(m.search_stream([q[:6], q[6:]], q_texts=[texts[:6], texts[6:]], top_k=10, aggregate=True))
    ok = True
    if rank == 0:
        single = MIPS(None, "in-memory", None, device=0, _store=conv(), rank=0, world=1, ivf=ivf)
        want = single.search(q, q_texts=texts, top_k=10, aggregate=True, agg_strat="opt1")
        want_v = single.search(q[:4], q_texts=texts[:4], top_k=5, return_idxs=True)
        for a, b in ((got, want), (got_v, want_v), (got_s[0] + got_s[1], want)):
            ok &= len(a) == len(b)
            for g, w in zip(a, b):
                ok &= len(g) == len(w)
                for x, y in zip(g, w):
                    ok &= all(x[k] == y[k] for k in ("doc_idx", "start_idx", "end_idx", "answer", "score"))
                    if y.get("start_vec") is not None:
                        ok &= bool(np.array_equal(x["start_vec"], y["start_vec"]) and np.array_equal(x["end_vec"], y["end_vec"]))
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
