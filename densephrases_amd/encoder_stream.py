"""Query encoder of batch t+1 against the search of batch t (reference order:
eval_phrase_retrieval.py:71-87 encodes a batch with ``query2vec`` and only then searches it, open_utils.py:83-101).

The encoder itself stays plain PyTorch-ROCm; this module is the plumbing that lets a serving loop
hand ``MIPS.search_stream`` device tensors that are still being computed: every ``next`` of an ``EncoderProducer`` launches
the encoder of one batch on a side stream and makes the stream the search is enqueued on wait for it."""
from __future__ import annotations


def encode_cls_pair(encs, ids, mask):
    """encoder.py:101-118: [B, T] token ids -> [B, 1536] = start [CLS] || end [CLS] (``query_start_encoder`` /
    ``query_end_encoder``), fp32, on the device."""
    import torch
    with torch.no_grad():
        s = encs[0](input_ids=ids, attention_mask=mask)[0][:, 0, :]
        e = encs[1](input_ids=ids, attention_mask=mask)[0][:, 0, :]
    return torch.cat([s, e], dim=1).float()


class EncoderProducer:
    """Iterable of device query tensors for ``MIPS.search_stream``: every ``next`` launches ``encode(ids, mask)`` of one batch
    on a side stream and makes the CURRENT stream (the one search_stream enqueues on) wait for it -- the batch already being
    searched was enqueued before, so its kernels and this encoder run side by side as far as the hardware lets them.
    ``side_stream=False`` runs the encoder on the search's own stream (the reference's order: one after the other)."""

    def __init__(self, encode, id_batches, masks, dev, side_stream=True):
        import torch
        self.encode, self.ids, self.masks, self.dev = encode, id_batches, masks, dev
        self.side = torch.cuda.Stream(device=dev) if (side_stream and dev.type == "cuda") else None

    def __iter__(self):
        import torch
        for ids, mask in zip(self.ids, self.masks):
            if self.side is None:
                yield self.encode(ids, mask)
                continue
            self.side.wait_stream(torch.cuda.current_stream(self.dev))     # the token ids were put there by the current stream
            with torch.cuda.stream(self.side):
                q = self.encode(ids, mask)
            torch.cuda.current_stream(self.dev).wait_stream(self.side)
            q.record_stream(torch.cuda.current_stream(self.dev))
            yield q


def make_bert_pair(dev, dtype, tiny=False):
    """Two random-init BERT-base encoders (no checkpoints offline; SpanBERT-base has this architecture): what
    ``DensePhrases.__init__`` loads as query_start_encoder / query_end_encoder (model.py:28-40)."""
    import torch
    from transformers import BertConfig, BertModel
    cfg = (BertConfig(hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=256) if tiny
           else BertConfig())                                             # bert-base: 12 layers, 768 hidden, 12 heads
    torch.manual_seed(0)
    return [BertModel(cfg, add_pooling_layer=False).to(device=dev, dtype=dtype).eval() for _ in range(2)]


def measure_overlap(mips, dev, B=512, T=64, k=20, steps=12, dtype=None, agg_strat="opt3"):
    """Encoder alone / search alone / one after the other on one stream / encoder on a side stream, ms per batch each, over
    ``mips`` (a MIPS over a resident shard).  Returns a dict; ``hidden`` = share of the encoder's time that disappeared behind
    the search."""
    import time
    import numpy as np
    import torch
    dtype = dtype or torch.bfloat16
    encs = make_bert_pair(dev, dtype)
    enc = lambda ids, mask: encode_cls_pair(encs, ids, mask)            # noqa: E731
    rng = np.random.default_rng(0)
    ids = [torch.from_numpy(rng.integers(1000, 29000, (B, T))).to(dev) for _ in range(steps + 2)]
    masks = [torch.ones_like(i) for i in ids]
    kw = dict(top_k=k, aggregate=True, agg_strat=agg_strat)

    def run(batches):
        t0 = time.perf_counter()
        n_out = sum(1 for _ in mips.search_stream(batches, **kw))
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n_out

    for i in range(2):
        enc(ids[i], masks[i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    qs = [enc(ids[i], masks[i]) for i in range(steps)]
    torch.cuda.synchronize()
    enc_ms = (time.perf_counter() - t0) / steps * 1e3
    # one untimed pass over ALL the batches first: the three timed runs below see the same queries (the encoders are deterministic),
    # so every document's metadata is in the host half's cache and no code object loads inside a timed run (round 4 timed the
    # search-alone run first and cold: it came out slower than encoder + search on one stream)
    run(qs)
    search_ms = run(qs) * 1e3
    serial_ms = run(EncoderProducer(enc, ids[:steps], masks[:steps], dev, side_stream=False)) * 1e3
    run(EncoderProducer(enc, ids[:2], masks[:2], dev))
    piped_ms = run(EncoderProducer(enc, ids[:steps], masks[:steps], dev)) * 1e3
    n_par = sum(p.numel() for e in encs for p in e.parameters())
    flop = 2.0 * n_par * B * T + 2 * 12 * 4.0 * B * T * T * 768            # weights + attention scores / mixes, both encoders
    del encs
    return {"encoder": f"2 x BERT-base ({str(dtype).split('.')[-1]}, random init), batch {B} x {T} tokens",
            "encoder_ms": enc_ms, "search_ms": search_ms, "serial_ms": serial_ms, "overlapped_ms": piped_ms,
            "encoder_tflops": flop / (enc_ms / 1e3) / 1e12,
            "queries_per_sec_overlapped": B / (piped_ms / 1e3), "queries_per_sec_serial": B / (serial_ms / 1e3),
            "queries_per_sec_search_only": B / (search_ms / 1e3), "steps": steps,
            "hidden": (enc_ms + search_ms - piped_ms) / enc_ms}
