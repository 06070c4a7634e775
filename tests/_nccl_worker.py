"""Worker of tests/test_dist_nccl.py (one process per GPU, launched by torch.distributed.run): range-sharded MIPS over
RCCL against the single-GPU MIPS on rank 0's device.  Exits non-zero on any mismatch."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    local = int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    from densephrases_amd import DocMeta, DocStore, MIPS
    from oracle.synth_dump import make_dump, make_queries
    docs = make_dump(seed=11, n_docs=300, d=768, n_par=4, words_per_par=(20, 40))
    conv = lambda: DocStore([DocMeta(m.doc_idx, m.title, m.context, m.f2o_start, m.word2char_start, m.word2char_end,  # noqa: E731
                                     m.start) for m in docs])
    m = MIPS(None, "in-memory", None, _store=conv())             # rank / world / device from the process group
    assert m.world == dist.get_world_size() and m.row_hi - m.row_lo < m.index.ntotal
    q = make_queries(np.random.default_rng(4), conv().rows, 12)
    texts = [f"q{i}" for i in range(12)]
    got = m.search(q, q_texts=texts, top_k=10, aggregate=True, agg_strat="opt1")
    got_v = m.search(q[:4], q_texts=texts[:4], top_k=5, return_idxs=True)
    ok = True
    if dist.get_rank() == 0:
        single = MIPS(None, "in-memory", None, device=local, _store=conv(), rank=0, world=1)
        want = single.search(q, q_texts=texts, top_k=10, aggregate=True, agg_strat="opt1")
        want_v = single.search(q[:4], q_texts=texts[:4], top_k=5, return_idxs=True)
        for a, b in ((got, want), (got_v, want_v)):
            for g, w in zip(a, b):
                ok &= len(g) == len(w)
                for x, y in zip(g, w):
                    ok &= all(x[k] == y[k] for k in ("doc_idx", "start_idx", "end_idx", "answer", "score"))
                    if y.get("start_vec") is not None:
                        ok &= bool(np.array_equal(x["start_vec"], y["start_vec"]) and np.array_equal(x["end_vec"], y["end_vec"]))
    flag = torch.tensor([1 if ok else 0], device=torch.device("cuda", local))
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
