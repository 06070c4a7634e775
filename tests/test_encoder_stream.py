"""densephrases_amd/encoder_stream.py on CPU (the GPU measurement is bench.py's also.encoder_overlap_b512)."""

def test_encoder_producer_hands_search_stream_one_tensor_per_batch_on_cpu():
    """densephrases_amd/encoder_stream.py (configs[4]: encoder of batch t+1 against the search of batch t): on a CPU device the
    producer degrades to calling the encoder in line -- one [B, 1536] fp32 tensor per batch, in order, start [CLS] || end [CLS]
    (encoder.py:101-118)."""
    import torch
    from densephrases_amd.encoder_stream import EncoderProducer, encode_cls_pair, make_bert_pair
    dev = torch.device("cpu")
    encs = make_bert_pair(dev, torch.float32, tiny=True)
    ids = [torch.randint(0, 30000, (3, 6)) for _ in range(4)]
    masks = [torch.ones_like(i) for i in ids]
    qs = list(EncoderProducer(lambda i, m: encode_cls_pair(encs, i, m), ids, masks, dev))
    assert len(qs) == 4 and all(q.shape == (3, 1536) and q.dtype == torch.float32 for q in qs)
    with torch.no_grad():
        want = torch.cat([encs[0](input_ids=ids[2], attention_mask=masks[2])[0][:, 0], encs[1](input_ids=ids[2], attention_mask=masks[2])[0][:, 0]], 1)
    assert torch.equal(qs[2], want)
