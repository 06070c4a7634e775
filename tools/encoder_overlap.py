#!/usr/bin/env python3
"""BASELINE configs[4] (KILT-style streaming, batch 512): the query encoder of batch t+1 against the search of batch t on ONE GPU.

The reference encodes a batch with two BERT-base forwards (encoder.py:101-118: ``query_start_encoder`` /
``query_end_encoder``, the [CLS] row of each) and only then searches it (eval_phrase_retrieval.py:71-87, one after the
other).  Here the encoder stays plain PyTorch-ROCm (random-init weights: no checkpoints offline) and
the question is what a serving loop pays for it next to ``MIPS.search_stream``:

    encoder alone        two BertModel forwards of [B, T] token ids -> [B, 1536] on the device, ms per batch
    search alone         MIPS.search_stream over resident query tensors, ms per batch
    pipelined            a producer that launches the encoder of batch t+1 on a side stream and hands search_stream the
                         device tensor (the search stream waits on the encoder's event): GPU work of the encoder competes with the
                         scan of batch t, the host half of batch t overlaps both

Prints one JSON line with the three times and ``hidden`` = (encoder + search - pipelined) / encoder, the share of the
encoder's time that disappeared behind the search.  The scan kernel owns every SIMD's register file while it runs
(DESIGN.md 5.1), so the expectation on one GPU is time slicing, not co-residency: this tool is the measurement.

    python tools/encoder_overlap.py [--rows 170000000] [--batch 512] [--tokens 64] [--dtype bf16] [--steps 6]
    python tools/encoder_overlap.py --dry        # CPU: tiny model, no shard -- checks the producer / stream logic only
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_encoders(dev, dtype, tiny=False):
    import torch
    from transformers import BertConfig, BertModel
    cfg = (BertConfig(hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=256) if tiny
           else BertConfig())                                             # bert-base: 12 layers, 768 hidden, 12 heads
    torch.manual_seed(0)
    encs = [BertModel(cfg, add_pooling_layer=False).to(device=dev, dtype=dtype).eval() for _ in range(2)]
    return encs


def encode(encs, ids, mask):
    """encoder.py:101-118: [B, T] ids -> [B, 1536] = start [CLS] || end [CLS], fp32, on the device"""
    import torch
    with torch.no_grad():
        s = encs[0](input_ids=ids, attention_mask=mask)[0][:, 0, :]
        e = encs[1](input_ids=ids, attention_mask=mask)[0][:, 0, :]
    return torch.cat([s, e], dim=1).float()


class EncoderProducer:
    """iterable of device query tensors for MIPS.search_stream: every ``next`` launches the encoder of one batch on a side
    stream and makes the CURRENT stream (the one search_stream enqueues on) wait for it -- the batch already being searched
    was enqueued before, so its kernels and this encoder run side by side as far as the hardware lets them"""

    def __init__(self, encs, id_batches, masks, dev, side_stream=True):
        import torch
        self.encs, self.ids, self.masks, self.dev = encs, id_batches, masks, dev
        self.side = torch.cuda.Stream(device=dev) if (side_stream and dev.type == "cuda") else None

    def __iter__(self):
        import torch
        for ids, mask in zip(self.ids, self.masks):
            if self.side is None:
                yield encode(self.encs, ids, mask)
                continue
            self.side.wait_stream(torch.cuda.current_stream(self.dev))     # the token ids were put there by the current stream
            with torch.cuda.stream(self.side):
                q = encode(self.encs, ids, mask)
            torch.cuda.current_stream(self.dev).wait_stream(self.side)
            q.record_stream(torch.cuda.current_stream(self.dev))
            yield q


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
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    if args.dry:
        dev = torch.device("cpu")
        encs = make_encoders(dev, torch.float32, tiny=True)
        ids = [torch.randint(0, 30000, (4, 8)) for _ in range(3)]
        masks = [torch.ones_like(i) for i in ids]
        qs = list(EncoderProducer(encs, ids, masks, dev))
        assert len(qs) == 3 and qs[0].shape == (4, 1536) and qs[0].dtype == torch.float32
        print(json.dumps({"dry": True, "batches": len(qs), "shape": list(qs[0].shape)}))
        return
    assert torch.cuda.is_available(), "needs a GPU (use --dry for the CPU check of the producer)"
    import __graft_entry__ as g
    g.build()
    from densephrases_amd import MIPS, Shard
    from densephrases_amd.synth import SynthDocStore
    dev = torch.device("cuda", 0)
    B, T, k, steps = args.batch, args.tokens, args.top_k, args.steps
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
    encs = make_encoders(dev, dtype)
    rng = np.random.default_rng(0)
    ids = [torch.from_numpy(rng.integers(1000, 29000, (B, T))).to(dev) for _ in range(steps + 2)]
    masks = [torch.ones_like(i) for i in ids]
    kw = dict(top_k=k, aggregate=True, agg_strat="opt3")

    def run(batches):
        t0 = time.perf_counter()
        n_out = sum(1 for _ in mips.search_stream(batches, **kw))
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n_out

    # encoder alone
    for i in range(2):
        encode(encs, ids[i], masks[i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    qs = [encode(encs, ids[i], masks[i]) for i in range(steps)]
    torch.cuda.synchronize()
    enc_ms = (time.perf_counter() - t0) / steps * 1e3
    # search alone over the encoded (resident) batches
    run(qs[:2])
    search_ms = run(qs) * 1e3
    # one after the other on one stream (the reference's order), then the side-stream producer
    serial_ms = run(EncoderProducer(encs, ids[:steps], masks[:steps], dev, side_stream=False)) * 1e3
    run(EncoderProducer(encs, ids[:2], masks[:2], dev))
    piped_ms = run(EncoderProducer(encs, ids[:steps], masks[:steps], dev)) * 1e3
    print(json.dumps({
        "workload": f"configs[4] shape on 1 GPU: batch {B} x {T} tokens, 2 x BERT-base ({args.dtype}, random init) -> "
                    f"MIPS.search_stream (top_k {k}, opt3) over {n} rows",
        "encoder_ms_per_batch": enc_ms, "search_ms_per_batch": search_ms,
        "encoder_then_search_same_stream_ms_per_batch": serial_ms, "encoder_on_side_stream_ms_per_batch": piped_ms,
        "queries_per_sec_pipelined": B / (piped_ms / 1e3), "queries_per_sec_search_only": B / (search_ms / 1e3),
        "hidden": (enc_ms + search_ms - piped_ms) / enc_ms}))


if __name__ == "__main__":
    main()
