"""Property tests (hypothesis) of the host-side logic around the hot path: shard partitioning, the exchange record,
the python halves of MIPS.search_phrase (paragraph / sentence cropping, aggregation) against the oracle's restatement
of the reference, and the synthetic-dump generator.  CPU only."""
import numpy as np
from hypothesis import assume, given, settings, strategies as st

from densephrases_amd.dist import RecordLayout, partition_rows
from densephrases_amd.index import MIPS, normalize_answer, split_sentences
from densephrases_amd.synth import synthetic_rows
from oracle import mips_oracle as O


@settings(max_examples=200, deadline=None)
@given(st.integers(0, 5_000_000), st.integers(1, 16), st.integers(1, 2000))
def test_partition_covers_contiguously_and_aligned(n, world, align):
    parts = partition_rows(n, world, align=align)
    assert len(parts) == world and parts[0][0] == 0 and parts[-1][1] == n
    for (a, b), (c, d) in zip(parts, parts[1:]):
        assert a <= b == c <= d
    for a, _ in parts[1:]:
        assert a % align == 0 or a == n


@settings(max_examples=100, deadline=None)
@given(st.lists(st.integers(1, 400), min_size=1, max_size=200), st.integers(1, 9))
def test_partition_cuts_fall_on_document_starts(doc_lens, world):
    starts = np.concatenate([[0], np.cumsum(doc_lens)[:-1]]).astype(np.int64)
    n = int(np.sum(doc_lens))
    parts = partition_rows(n, world, doc_starts=starts)
    assert parts[0][0] == 0 and parts[-1][1] == n
    ok = set(starts.tolist()) | {n}
    for a, b in parts:
        assert a in ok and b in ok and a <= b


@settings(max_examples=50, deadline=None)
@given(st.integers(1, 300), st.integers(1, 64), st.integers(1, 5))
def test_record_layout_fields_are_disjoint_and_aligned(n, k, world):
    import torch
    lay = RecordLayout(n, k)
    spans = sorted(lay.fields.values())
    for (o0, b0), (o1, _) in zip(spans, spans[1:]):
        assert o0 % 8 == 0 and o0 + b0 <= o1
    assert spans[-1][0] + spans[-1][1] <= lay.nbytes and lay.nbytes % 8 == 0
    buf = torch.zeros((world, lay.nbytes), dtype=torch.uint8)
    v = lay.views(buf)
    assert v["D"].shape == (world, n, k) and v["status"].shape == (world, n) and v["bound"].shape == (world, n)
    v["bound"][world - 1, n - 1] = -1e300
    v["I"][0, 0, 0] = -1
    assert float(lay.views(buf)["bound"][world - 1, n - 1]) == -1e300 and int(lay.views(buf)["I"][0, 0, 0]) == -1


_words = st.lists(st.sampled_from(["alpha", "Bravo.", "c!", "delta?", "e.g.", "Mr.", "x", "[PAR]", "end."]),
                  min_size=1, max_size=40)


@settings(max_examples=200, deadline=None)
@given(_words)
def test_split_sentences_partitions_the_text(words):
    text = " ".join(words)
    sents = split_sentences(text)
    assert sents and sents[0][1] == 0
    for (s, off), nxt in zip(sents, sents[1:] + [(None, len(text))]):
        assert text[off:off + len(s)] == s
        assert text[off + len(s):nxt[1]].strip() == ""          # only whitespace between sentences
    assert [s for s, _ in sents] == [s for s, _ in O.rule_sentences(text)]      # same rule as the oracle's restatement


@settings(max_examples=200, deadline=None)
@given(st.lists(st.lists(st.sampled_from(["a", "bb", "ccc", "dd.", "e"]), min_size=1, max_size=8), min_size=1,
                max_size=6), st.data())
def test_adjust_keeps_the_answer_and_matches_the_oracle(paragraphs, data):
    ctx = " [PAR] ".join(" ".join(p) for p in paragraphs)
    pi = data.draw(st.integers(0, len(paragraphs) - 1))
    par_start = len(" [PAR] ".join(" ".join(p) for p in paragraphs[:pi])) + (len(" [PAR] ") if pi else 0)
    par_len = len(" ".join(paragraphs[pi]))
    a = data.draw(st.integers(par_start, par_start + par_len - 1))
    b = data.draw(st.integers(a + 1, par_start + par_len))
    assume(not ctx[a].isspace() and not ctx[b - 1].isspace())      # phrase spans start and end on a token
    each = {"context": ctx, "start_pos": a, "end_pos": b}
    got = MIPS.adjust(dict(each))
    assert got["context"] == " ".join(paragraphs[pi])
    assert got["context"][got["start_pos"]:got["end_pos"]] == ctx[a:b]
    assert got == O.adjust(dict(each))
    sent = MIPS.adjust_sent(dict(got))
    assert sent["context"][sent["start_pos"]:sent["end_pos"]] == ctx[a:b]
    assert sent == O.adjust_sent(dict(got))


@settings(max_examples=100, deadline=None)
@given(st.text(alphabet="aA theTHE,.!? x1", max_size=40))
def test_normalize_answer_is_idempotent(s):
    assert normalize_answer(normalize_answer(s)) == normalize_answer(s)


@settings(max_examples=100, deadline=None)
@given(st.lists(st.tuples(st.sampled_from(["T1", "T2", "T3"]), st.integers(0, 3), st.integers(4, 6),
                          st.sampled_from(["ctx a", "ctx b"]), st.sampled_from(["The Answer", "answer", "other"]),
                          st.floats(-50, 50, allow_nan=False)), min_size=0, max_size=12),
       st.sampled_from(["opt1", "opt2", "opt3", "opt4"]))
def test_aggregate_results_matches_the_oracle(rows, strat):
    def mk():
        return [{"title": [t], "start_pos": s, "end_pos": e, "context": c, "answer": a, "score": sc}
                for t, s, e, c, a, sc in sorted(rows, key=lambda r: -r[5])]
    m = MIPS.__new__(MIPS)
    got = m.aggregate_results(mk(), 10, "q", strat)
    want = O.aggregate_results(mk(), strat)
    assert got == want


@settings(max_examples=30, deadline=None)
@given(st.integers(0, 10_000_000), st.integers(1, 40), st.integers(1, 39), st.integers(0, 2**31))
def test_synthetic_rows_do_not_depend_on_chunking(row0, n, cut, seed):
    cut = min(cut, n)
    whole = synthetic_rows(row0, n, seed)
    assert whole.dtype == np.int8 and whole.shape == (n, 768)
    np.testing.assert_array_equal(whole[:cut], synthetic_rows(row0, cut, seed))
    np.testing.assert_array_equal(whole[cut:], synthetic_rows(row0 + cut, n - cut, seed))
