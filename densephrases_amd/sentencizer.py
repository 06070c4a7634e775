"""Sentence units for ``return_sent`` (index.py:65-66, 178-187): the reference runs spaCy 2.3's ``English()`` tokenizer and
the rule-based ``sentencizer`` pipe over the cropped context and takes ``(X.text, X[0].idx)`` of every sentence span.

spaCy cannot be installed here and nothing in the tree pins it, so this is a RESTATEMENT FROM MEMORY of the part of both
that decides sentence boundaries -- closer than "split after . ! ? + whitespace", still approximate:

tokenizer (spacy/tokenizer.pyx, lang/punctuation.py, lang/en/tokenizer_exceptions.py)
    * text is cut at whitespace; a chunk that is a *special case* (``Mr.``, ``U.S.``, ``e.g.``, ``a.m.``, months, states, a
      single lower-case letter + '.', ...) is one token;
    * otherwise prefixes and suffixes are peeled one at a time, re-checking the remainder against the special cases:
      prefixes = punctuation / quotes / brackets / currency / ellipsis; suffixes = the same punctuation and quotes,
      ellipsis (``..+`` is ONE token), ``'s``, and a final '.' only when it follows a lower-case letter, a digit, ``%²-+``,
      a quote or punctuation, or two upper-case letters (``J.`` and ``U.S.`` keep their period);
    * inside the remainder an ellipsis, and a '.' between a lower-case letter or quote and an upper-case letter or quote, are
      infixes (``end.Next`` -> 3 tokens); the pieces between infixes are not split further.
sentencizer (spacy/pipeline/pipes.pyx, Sentencizer.predict)
    * a token whose text is one of ``punct_chars`` arms the split; the next token that is neither punctuation (every
      character in a Unicode P* category) nor in ``punct_chars`` starts a sentence -- so closing AND opening quotes or
      brackets after a full stop stay with the sentence that ended, and an ellipsis does not end one.

whitespace (Tokenizer.__call__)
    * one blank after a token belongs to that token; anything more in a run of whitespace (and a run at the start of the
      text) is a token that can start a sentence: ``End.  Next`` -> the second sentence starts at the second blank.

Not modelled: the emoticon / URL token matches, unit and currency suffixes after digits, the remaining infixes (none of
them yields a ``punct_chars`` token).  densephrases_amd/csrc/dph_host.cpp carries the same rule; tests/test_host_half_golden.py holds the two against
each other.
"""
from __future__ import annotations

import unicodedata

# the BMP part of spaCy 2.3's Sentencizer.default_punct_chars
PUNCT_CHARS = frozenset(
    "!.?։؟۔܀܁܂߹।॥၊။።፧፨᙮᜵᜶᠃᠉"
    "᥄᥅᪨᪩᪪᪫᭚᭛᭞᭟᰻᰼᱾᱿‼‽⁇⁈⁉⸮"
    "⸼꓿꘎꘏꛳꛷꡶꡷꣎꣏꤯꧈꧉꩝꩞꩟꫰꫱꯫﹒"
    "﹖﹗！．？｡。")
# lang/char_classes.py: _punct and _quotes (single characters)
PUNCT_CLASS = frozenset("…,:;!?¿؟¡()[]{}<>_#*&。？！，、；：～·।،۔؛٪")
QUOTES = frozenset("'\"”“`‘´’‚,„»«「」『』（）〔〕【】《》〈〉")
_PREFIX = PUNCT_CLASS | QUOTES | frozenset("§%=—–$£€¥฿₽﷼₴")
_SUFFIX = PUNCT_CLASS | QUOTES | frozenset("—–")
_BEFORE_PERIOD = PUNCT_CLASS | QUOTES | frozenset("%²-+")
# lang/en/tokenizer_exceptions.py (abbreviations that end in a period) and lang/tokenizer_exceptions.py (BASE_EXCEPTIONS)
ABBREVIATIONS = frozenset("""
a.m. p.m. Adm. Bros. co. Co. Corp. D.C. Dr. e.g. E.g. E.G. Gen. Gov. i.e. I.e. I.E. Inc. Jr. Ltd. Md. Messrs. Mo. Mont. Mr.
Mrs. Ms. Ph.D. Prof. Rep. Rev. Sen. St. vs. v.s. Mt. Ak. Ala. Apr. Ariz. Ark. Aug. Calif. Colo. Conn. Dec. Del. Feb. Fla.
Ga. Ia. Id. Ill. Ind. Jan. Jul. Jun. Kan. Kans. Ky. La. Mar. Mass. Mich. Minn. Miss. N.C. N.D. N.H. N.J. N.M. N.Y. Neb.
Nebr. Nev. Nov. Oct. Okla. Ore. Pa. S.C. Sep. Sept. Tenn. Va. Wash. Wis.
""".split())

TERM, PUNCT, OTHER = 0, 1, 2


def is_punct_char(c: str) -> bool:
    return unicodedata.category(c)[0] == "P"


def _special(text: str, a: int, b: int) -> bool:
    if b - a == 2 and text[a + 1] == "." and "a" <= text[a] <= "z":
        return True                                  # BASE_EXCEPTIONS: "a." .. "z."
    if 5 <= b - a <= 6 and text[b - 4:b] in ("a.m.", "p.m."):
        h = text[a:b - 4]                            # "1a.m." .. "12p.m." (two tokens in spaCy, neither a full stop)
        return h.isascii() and h.isdigit() and h[0] != "0" and 1 <= int(h) <= 12
    return text[a:b] in ABBREVIATIONS


def _kind(text: str, a: int, b: int) -> int:
    if b - a == 1 and text[a] in PUNCT_CHARS:
        return TERM
    return PUNCT if all(is_punct_char(c) for c in text[a:b]) else OTHER


def _prefix_len(text: str, a: int, b: int) -> int:
    """length of the prefix the tokenizer would cut off text[a:b] (0 = none)"""
    c = text[a]
    if c == ".":
        j = a
        while j < b and text[j] == ".":
            j += 1
        return j - a if j - a >= 2 else 0            # ellipsis
    if c in _PREFIX:
        return 1
    if c == "+" and not (a + 1 < b and "0" <= text[a + 1] <= "9"):
        return 1
    return 0


def _suffix_len(text: str, a: int, b: int) -> int:
    """length of the suffix the tokenizer would cut off text[a:b] (0 = none)"""
    c = text[b - 1]
    if c == ".":
        k = b
        while k > a and text[k - 1] == ".":
            k -= 1
        if b - k >= 2:
            return b - k                             # ellipsis: ONE token
        if b - 2 >= a:
            p = text[b - 2]
            if p.islower() or "0" <= p <= "9" or p in _BEFORE_PERIOD or (b - 3 >= a and p.isupper() and text[b - 3].isupper()):
                return 1
        return 0
    if c in _SUFFIX:
        return 1
    if b - a >= 2 and c in "sS" and text[b - 2] in "'\u2019":
        return 2
    return 0


def _tokens_of_chunk(text: str, a: int, b: int, out: list):
    """tokens (start, end, kind) of the whitespace-free chunk text[a:b], in text order (Tokenizer._split_affixes, then the
    infixes that matter for sentence boundaries)"""
    tail = []                                        # suffix tokens, outermost first
    while a < b and not _special(text, a, b):
        pre = _prefix_len(text, a, b)
        if pre and a + pre < b and _special(text, a + pre, b):
            out.append((a, a + pre, _kind(text, a, a + pre))); a += pre
            break
        suf = _suffix_len(text, a, b)
        if suf and a < b - suf and _special(text, a, b - suf):
            tail.append((b - suf, b, _kind(text, b - suf, b))); b -= suf
            break
        if pre and suf and pre + suf <= b - a:
            out.append((a, a + pre, _kind(text, a, a + pre))); a += pre
            tail.append((b - suf, b, _kind(text, b - suf, b))); b -= suf
        elif pre:
            out.append((a, a + pre, _kind(text, a, a + pre))); a += pre
        elif suf:
            tail.append((b - suf, b, _kind(text, b - suf, b))); b -= suf
        else:
            break
    if a < b:
        if _special(text, a, b):
            out.append((a, b, OTHER))
        else:
            # infixes (Tokenizer._attach_tokens): an ellipsis ("..+" or the single character), or a '.' between [lower-case or
            # quote] and [upper-case or quote]; a match at the start of the piece being collected is skipped, as spaCy does
            s = i = a
            while i < b:
                n = 0
                if text[i] == ".":
                    j = i
                    while j < b and text[j] == ".":
                        j += 1
                    if j - i >= 2:
                        n = j - i
                    elif (i > a and i + 1 < b and (text[i - 1].islower() or text[i - 1] in QUOTES)
                          and (text[i + 1].isupper() or text[i + 1] in QUOTES)):
                        n = 1
                elif text[i] == "\u2026":
                    n = 1
                if n == 0:
                    i += 1
                    continue
                if i > s:
                    out.append((s, i, _kind(text, s, i)))
                    out.append((i, i + n, _kind(text, i, i + n)))
                    s = i + n
                i += n
            if s < b:
                out.append((s, b, _kind(text, s, b)))
    out.extend(reversed(tail))


def tokenize(text: str):
    """tokens (start, end, kind) of the text.  Whitespace (Tokenizer.__call__): ONE blank after a token belongs to that token;
    whatever else a run of whitespace holds -- and a run at the very start of the text -- is a token of its own, which is
    neither punctuation nor a full stop, so it can start a sentence."""
    out, i, n = [], 0, len(text)
    while i < n:
        j = i
        if text[i].isspace():
            while j < n and text[j].isspace():
                j += 1
            a = i + 1 if (i > 0 and text[i] == " ") else i
            if a < j:
                out.append((a, j, OTHER))
        else:
            while j < n and not text[j].isspace():
                j += 1
            _tokens_of_chunk(text, i, j, out)
        i = j
    return out


def split_sentences(text: str):
    """[(sentence_text, start_char)] as the reference reads them off ``self.sentencizer(context).sents`` (index.py:179)"""
    toks = tokenize(text)
    if not toks:
        return []
    starts = [0]                                     # token index of every sentence start
    seen = False
    for t, (a, b, kind) in enumerate(toks):
        if seen and kind == OTHER:
            starts.append(t)
            seen = False
        elif kind == TERM:
            seen = True
    out = []
    for s, t0 in enumerate(starts):
        t1 = starts[s + 1] if s + 1 < len(starts) else len(toks)
        out.append((text[toks[t0][0]:toks[t1 - 1][1]], toks[t0][0]))
    return out
