"""Sentence units of ``return_sent`` (index.py:65-66, 178-187 run spaCy 2.3's English tokenizer + rule-based sentencizer).
spaCy is absent offline, so nothing here is pinned against it: these tests hold THREE independently written restatements
of the same published rules against each other -- oracle/spacy_sentencizer.py (regular expressions compiled the way spaCy
compiles them), densephrases_amd/sentencizer.py (code-point walk, the python twin of the product rule) and
csrc/dph_host.cpp (what MIPS runs) -- and against a table of behaviours worked out by hand from those rules."""
import importlib.util
import os
import random

import pytest

from oracle import spacy_sentencizer as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _three(text):
    from densephrases_amd import _dph_host
    from densephrases_amd.sentencizer import split_sentences
    py = split_sentences(text)
    cpp = [(text[a:b], a) for a, b in _dph_host.split_sentences(text)]
    return py, cpp, O.sentences(text)


# (text, [sentence texts]) -- what English() + sentencizer yield by the rules restated in oracle/spacy_sentencizer.py
BEHAVIOUR = [
    ("One two. Three! Four? five", ["One two.", "Three!", "Four?", "five"]),
    # whitespace: one blank belongs to the token before it; the rest of a run is a token, and it can start a sentence
    ("One two. Three!  Four? five", ["One two.", "Three!", " Four?", "five"]),
    ("End.\nNext line.\n", ["End.", "\nNext line.", "\n"]),
    # special cases keep their period: no full-stop token, no split
    ("He met Mr. Smith in the U.S. Army. It was 3.5 km away.", ["He met Mr. Smith in the U.S. Army.", "It was 3.5 km away."]),
    ("Born in St. Louis, Mo. in 1900. Died at 5p.m. today!", ["Born in St. Louis, Mo. in 1900.", "Died at 5p.m. today!"]),
    # the period rule: off after a lower-case letter / digit / quote / punctuation / two capitals, kept after ONE capital
    ("J. K. Rowling wrote it. AB. Then c.", ["J. K. Rowling wrote it.", "AB.", "Then c."]),
    ("It cost 5. (Yes). Fine.", ["It cost 5. (", "Yes).", "Fine."]),
    # closing quotes stay with the sentence that ended ...
    ('She said "Go." Then left.', ['She said "Go."', "Then left."]),
    # ... and so do OPENING quotes / brackets after a full stop: the next sentence starts at the first non-punctuation token
    ('He asked. "Why?" she said.', ['He asked. "', 'Why?"', "she said."]),
    ("end. [PAR] next", ["end. [", "PAR] next"]),
    # an ellipsis is one token that is not in punct_chars
    ("Wait... what? Yes.", ["Wait... what?", "Yes."]),
    ("Wait… what", ["Wait… what"]),
    # doubled terminators are separate tokens; both arm the split
    ("Is it?! Yes.", ["Is it?!", "Yes."]),
    # infix: [lower|quote] '.' [upper|quote]
    ("the end.Next one", ["the end.", "Next one"]),
    ('he said."Then left', ["he said.", '"Then left']),
    ("version 3.5.Next", ["version 3.5.Next"]),
    # other scripts' terminators from Sentencizer.default_punct_chars
    ("日本語。 次の文！ Fin.", ["日本語。", "次の文！", "Fin."]),
    ("क्या है। नहीं", ["क्या है।", "नहीं"]),
    # a symbol is not punctuation (category S*): it starts the sentence
    ("Paid. $5 each", ["Paid.", "$5 each"]),
    ("  leading. trailing  ", ["  leading.", "trailing  "]),
    ("leading. trailing ", ["leading.", "trailing"]),
    ("no terminator", ["no terminator"]),
    (" ", [" "]),
]


@pytest.mark.parametrize("text,want", BEHAVIOUR)
def test_behaviour_table(text, want):
    py, cpp, orc = _three(text)
    assert [t for t, _ in orc] == want
    assert py == orc and cpp == orc
    for t, s in orc:
        assert text[s:s + len(t)] == t


def test_empty_text_has_no_sentences():
    assert _three("") == ([], [], [])


def test_three_restatements_agree_on_generated_text():
    rng = random.Random(5)
    vocab = ["alpha", "Bravo", "c", "Delta", "e.g.", "Mr.", "Mrs.", "U.S.", "x", "[PAR]", "end", "(see", "below)", '"Quote', 'quote"',
             "'s", "it's", "5p.m.", "12a.m.", "13a.m.", "3.5", "J.", "K.", "St.", "Louis", "...", "…", "?!", "??", "wait...what",
             "end.Next", 'said."Then', "$5.50", "+1", "+x", "a.", "z.", "Ph.D.", "Inc.", "。", "日本語。", "次の文！", "¿Qué?", "—", "–",
             "co.", "N.Y.", "(e.g.", "i.e.)", "[1]", "«a»", "»", "x².", "", " ", "\n", "'", "''", '"', "U.S.A.", "A.B", "AB.",
             "Ab.", "aB.", "Ünï.", "straße.", "ÉCOLE.", "ß."]
    for _ in range(4000):
        text = " ".join(rng.choice(vocab) + rng.choice([".", "!", "?", "", ""]) + rng.choice(["", "", '"', ")", "'"])
                        for _ in range(rng.randint(1, 12)))
        py, cpp, orc = _three(text)
        assert py == orc and cpp == orc, text
    alphabet = list("abAB19 .  .!?\"'()[]…,;:-+%$s’“”«»。！²\n\t")
    for _ in range(40000):
        text = "".join(rng.choice(alphabet) for _ in range(rng.randint(0, 14)))
        py, cpp, orc = _three(text)
        assert py == orc and cpp == orc, repr(text)


def test_sentences_partition_the_text_up_to_whitespace():
    rng = random.Random(9)
    words = ["alpha", "Bravo.", "c!", "delta?", "e.g.", "Mr.", '"x"', "(y).", "end.Next", "...", "", "\n"]
    for _ in range(500):
        text = " ".join(rng.choice(words) for _ in range(rng.randint(1, 30)))
        if not text:
            continue
        sents = O.sentences(text)
        assert sents[0][1] == 0
        for (s, off), nxt in zip(sents, sents[1:] + [(None, len(text))]):
            assert text[off:off + len(s)] == s and text[off + len(s):nxt[1]].strip() == ""


def test_generated_tables_are_current():
    """csrc/dph_sentencizer_tables.inc and csrc/dph_unicode_punct.inc are what tools/gen_sentencizer_tables.py writes from
    densephrases_amd/sentencizer.py and this interpreter's unicodedata"""
    spec = importlib.util.spec_from_file_location("_gen", os.path.join(ROOT, "tools", "gen_sentencizer_tables.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    for name, text in (("dph_unicode_punct.inc", gen.punct_inc()), ("dph_sentencizer_tables.inc", gen.tables_inc())):
        with open(os.path.join(ROOT, "densephrases_amd", "csrc", name)) as f:
            assert f.read() == text, f"{name} is stale: run python tools/gen_sentencizer_tables.py"


def test_the_oracle_and_the_product_list_the_same_special_cases():
    from densephrases_amd import sentencizer as S
    letters = {c + "." for c in "abcdefghijklmnopqrstuvwxyz"}
    hours = {f"{h}{p}" for h in range(1, 13) for p in ("a.m.", "p.m.")}
    assert O._SPECIALS - letters - hours == set(S.ABBREVIATIONS)
    assert O.PUNCT_CHARS == set(S.PUNCT_CHARS)
