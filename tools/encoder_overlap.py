#!/usr/bin/env python3
"""BASELINE configs[4] (KILT-style streaming, batch 512): the query encoder of batch t+1 against the search of batch t on ONE GPU.

The reference encodes a batch with two BERT-base forwards (encoder.py:101-118: ``query_start_encoder`` /
``query_end_encoder``, the [CLS] row of each) and only then searches it (eval_phrase_retrieval.py:71-87, one after the
other).  Here the encoder stays plain PyTorch-ROCm (random-init weights: no checkpoints offline) and
the question is what a serving loop pays for it next to ``MIPS.search_stream`` (densephrases_amd/encoder_stream.py:
``measure_overlap``; bench.py reports the same measurement as ``also.encoder_overlap_b512``):

    encoder alone        two BertModel forwards of [B, T] token ids -> [B, 1536] on the device, ms per batch
    search alone         MIPS.search_stream over resident query tensors, ms per batch
    serial               encoder and search on one stream, the reference's order
    overlapped           a producer that launches the encoder of batch t+1 on a side stream and hands search_stream the
                         device tensor (the search stream waits on the encoder's event)

    python tools/encoder_overlap.py [--rows 170000000] [--batch 512] [--tokens 64] [--dtype bf16] [--steps 6]
    python tools/encoder_overlap.py --dry        # CPU: tiny model, no shard -- checks the producer / stream logic only
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=170_000_000)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--tokens", type=int, default=64, help="max_query_length 64 (options.py); KILT runs use up to 384")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--top_k", type=int, default=20, help="retrieval_unit='document': 2 * top_k (model.py:79-81)")
    ap.add_argument("--dry", action="store_true")
    args = ap.parse_args()
    import torch
    from densephrases_amd.encoder_stream import EncoderProducer, encode_cls_pair, make_bert_pair, measure_overlap
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    if args.dry:
        dev = torch.device("cpu")
        encs = make_bert_pair(dev, torch.float32, tiny=True)
        ids = [torch.randint(0, 30000, (4, 8)) for _ in range(3)]
        masks = [torch.ones_like(i) for i in ids]
        qs = list(EncoderProducer(lambda i, m: encode_cls_pair(encs, i, m), ids, masks, dev))
        assert len(qs) == 3 and qs[0].shape == (4, 1536) and qs[0].dtype == torch.float32
        print(json.dumps({"dry": True, "batches": len(qs), "shape": list(qs[0].shape)}))
        return
    assert torch.cuda.is_available(), "needs a GPU (use --dry for the CPU check of the producer)"
    import __graft_entry__ as g
    g.build()
    from densephrases_amd import MIPS, Shard
    from densephrases_amd.synth import SynthDocStore
    dev = torch.device("cuda", 0)
    n = args.rows // 32 * 32
    shard = Shard(n, device=0)
    shard.fill_synthetic(seed=42, kind=0)
    doc = (np.arange(n, dtype=np.int64) // 100).astype(np.int32)
    word = (np.arange(n, dtype=np.int64) % 100).astype(np.int32)
    shard.set_idx2id(doc, word)
    nd = (n + 99) // 100
    shard.set_f2o(np.arange(nd + 1, dtype=np.int32), np.arange(0, (nd + 2) * 100, 100, dtype=np.int64),
                  np.tile(np.arange(100, dtype=np.int32), nd + 1))
    del doc, word
    shard.finalize()
    mips = MIPS.from_shard(shard, SynthDocStore())
    out = measure_overlap(mips, dev, B=args.batch, T=args.tokens, k=args.top_k, steps=args.steps, dtype=dtype)
    out["workload"] = f"configs[4] shape on 1 GPU: {out['encoder']} -> MIPS.search_stream (top_k {args.top_k}, opt3) over {n} rows"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
