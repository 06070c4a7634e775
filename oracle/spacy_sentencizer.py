"""TEST INFRASTRUCTURE (oracle): the sentence units of ``return_sent`` as the reference obtains them -- spaCy 2.3's
``English()`` tokenizer followed by the rule-based ``sentencizer`` pipe (/root/reference/densephrases/index.py:65-66,
178-187: ``[(X.text, X[0].idx) for X in self.sentencizer(context).sents]``).

spaCy is a third-party dependency that is absent here (``spacy==2.3.2`` in the reference's requirements; not installable
offline), so this restates its published algorithm FROM MEMORY in spaCy's own shape -- prefix / suffix / infix regular
expressions compiled from the character classes of ``spacy/lang/char_classes.py`` and ``spacy/lang/punctuation.py``, the
``_split_affixes`` loop of ``spacy/tokenizer.pyx`` with its special-case checks, the abbreviation special cases of
``spacy/lang/en/tokenizer_exceptions.py`` and ``spacy/lang/tokenizer_exceptions.py``, and ``Sentencizer.predict`` of
``spacy/pipeline/pipes.pyx``.  **Parity unpinned**: no spaCy output exists in the tree to check it against.

Deliberately written differently from the product's splitter (densephrases_amd/sentencizer.py and csrc/dph_host.cpp walk
code-point indices with hand-written predicates; this module compiles regular expressions and slices strings), so that the
property tests comparing the two catch slips in either.

Left out (none of them produces a sentence-final token): emoticons and URL token matches, icons, multi-character currency
symbols, unit suffixes after digits, the contraction special cases (``don't`` -> ``do`` ``n't``).
"""
import re
import unicodedata

_punct = "… …… , : ; ! ? ¿ ؟ ¡ ( ) [ ] { } < > _ # * & 。 ？ ！ ， 、 ； ： ～ · । ، ۔ ؛ ٪"
_quotes = "' \" ” “ ` ‘ ´ ’ ‚ , „ » « 「 」 『 』 （ ） 〔 〕 【 】 《 》 〈 〉"
_currency = "$ £ € ¥ ฿ ₽ ﷼ ₴"

LIST_PUNCT = [re.escape(p) for p in _punct.split()]
LIST_QUOTES = [re.escape(q) for q in _quotes.split()]
LIST_ELLIPSES = [r"\.\.+", "…"]
LIST_CURRENCY = [re.escape(c) for c in _currency.split()]
PUNCT = "|".join(LIST_PUNCT)
CONCAT_QUOTES = "".join(re.escape(q) for q in _quotes.split())
# every cased BMP letter, like spaCy's ALPHA_LOWER / ALPHA_UPPER character classes (which enumerate Unicode blocks)
ALPHA_LOWER = "".join(chr(c) for c in range(0x10000) if chr(c).islower())
ALPHA_UPPER = "".join(chr(c) for c in range(0x10000) if chr(c).isupper())

_prefixes = ["§", "%", "=", "—", "–", r"\+(?![0-9])"] + LIST_PUNCT + LIST_ELLIPSES + LIST_QUOTES + LIST_CURRENCY
# spaCy: r"(?<=[0-9{al}{e}{p}(?:{q})])\." -- a final period comes off after a digit, a lower-case letter, one of %²-+, punctuation
# or a quote (single characters, except the double ellipsis, which a look-behind on its last character covers), and
# r"(?<=[{au}][{au}])\." -- after two upper-case letters
_before_period = "0-9" + ALPHA_LOWER + r"%²\-\+" + CONCAT_QUOTES + "".join(re.escape(p) for p in _punct.split() if len(p) == 1)
_suffixes = (LIST_PUNCT + LIST_ELLIPSES + LIST_QUOTES + ["'s", "'S", "’s", "’S", "—", "–"]
             + [r"(?<=[{b}])\.".format(b=_before_period), r"(?<=[{au}][{au}])\.".format(au=ALPHA_UPPER)])
_infixes = LIST_ELLIPSES + [r"(?<=[{al}{q}])\.(?=[{au}{q}])".format(al=ALPHA_LOWER, au=ALPHA_UPPER, q=CONCAT_QUOTES)]

prefix_search = re.compile("|".join("^" + p for p in _prefixes)).search
suffix_search = re.compile("|".join(s + "$" for s in _suffixes)).search
infix_finditer = re.compile("|".join(_infixes)).finditer

# abbreviations that end in a period and stay one token
_SPECIALS = set("""
a.m. p.m. Adm. Bros. co. Co. Corp. D.C. Dr. e.g. E.g. E.G. Gen. Gov. i.e. I.e. I.E. Inc. Jr. Ltd. Md. Messrs. Mo. Mont. Mr.
Mrs. Ms. Ph.D. Prof. Rep. Rev. Sen. St. vs. v.s. Mt. Ak. Ala. Apr. Ariz. Ark. Aug. Calif. Colo. Conn. Dec. Del. Feb. Fla.
Ga. Ia. Id. Ill. Ind. Jan. Jul. Jun. Kan. Kans. Ky. La. Mar. Mass. Mich. Minn. Miss. N.C. N.D. N.H. N.J. N.M. N.Y. Neb.
Nebr. Nev. Nov. Oct. Okla. Ore. Pa. S.C. Sep. Sept. Tenn. Va. Wash. Wis.
""".split())
_SPECIALS |= {c + "." for c in "abcdefghijklmnopqrstuvwxyz"}                       # BASE_EXCEPTIONS
_SPECIALS |= {f"{h}{p}" for h in range(1, 13) for p in ("a.m.", "p.m.")}            # "1a.m." .. "12p.m."

# Sentencizer.default_punct_chars (the BMP part)
PUNCT_CHARS = set("!.?։؟۔܀܁܂߹।॥၊။።፧፨᙮᜵᜶᠃᠉᥄᥅᪨᪩᪪᪫᭚᭛᭞᭟᰻᰼᱾᱿"
                  "‼‽⁇⁈⁉⸮⸼꓿꘎꘏꛳꛷꡶꡷꣎꣏꤯꧈꧉꩝꩞꩟꫰꫱꯫﹒﹖﹗！．？｡。")


def is_punct(text):
    """spacy.lang.lex_attrs.is_punct"""
    return all(unicodedata.category(c).startswith("P") for c in text)


def _split_affixes(string):
    """Tokenizer._split_affixes: (prefixes, remainder, suffixes innermost-last)"""
    prefixes, suffixes = [], []
    last_size = 0
    while string and len(string) != last_size:
        if string in _SPECIALS:
            break
        last_size = len(string)
        m = prefix_search(string)
        pre_len = m.end() if m else 0
        if pre_len:
            minus_pre = string[pre_len:]
            if minus_pre and minus_pre in _SPECIALS:
                prefixes.append(string[:pre_len])
                string = minus_pre
                break
        m = suffix_search(string)
        suf_len = len(string) - m.start() if m else 0
        if suf_len:
            minus_suf = string[:-suf_len]
            if minus_suf and minus_suf in _SPECIALS:
                suffixes.append(string[-suf_len:])
                string = minus_suf
                break
        if pre_len and suf_len and pre_len + suf_len <= len(string):
            prefixes.append(string[:pre_len])
            suffixes.append(string[-suf_len:])
            string = string[pre_len:-suf_len]
        elif pre_len:
            prefixes.append(string[:pre_len])
            string = string[pre_len:]
        elif suf_len:
            suffixes.append(string[-suf_len:])
            string = string[:-suf_len]
    return prefixes, string, suffixes


def _chunk_tokens(chunk):
    """token strings of a whitespace-free chunk, in order (Tokenizer._tokenize + _attach_tokens)"""
    prefixes, string, suffixes = _split_affixes(chunk)
    toks = list(prefixes)
    if string:
        if string in _SPECIALS:
            toks.append(string)
        else:
            start = 0
            for m in infix_finditer(string):
                if m.start() == start:              # (as in spaCy: an infix where the current piece starts is not split off)
                    continue
                toks.append(string[start:m.start()])
                if m.start() != m.end():
                    toks.append(m.group())
                start = m.end()
            if start < len(string):
                toks.append(string[start:])
    toks.extend(reversed(suffixes))
    return toks


def tokenize(text):
    """[(token_text, idx)] (Tokenizer.__call__): non-whitespace chunks are split by the rules above; of a run of whitespace a
    single leading blank is the trailing space of the token before it, the rest (all of it at the start of the text, or
    when it does not begin with a blank) is a whitespace token"""
    out = []
    for m in re.finditer(r"\s+|\S+", text):
        chunk, pos = m.group(), m.start()
        if chunk[0].isspace():
            if pos > 0 and chunk[0] == " ":
                chunk, pos = chunk[1:], pos + 1
            if chunk:
                out.append((chunk, pos))
            continue
        for t in _chunk_tokens(chunk):
            out.append((t, pos))
            pos += len(t)
        assert pos == m.end()
    return out


def sentences(text):
    """[(X.text, X[0].idx) for X in doc.sents] of ``English()`` + ``sentencizer`` (Sentencizer.predict)"""
    toks = tokenize(text)
    if not toks:
        return []
    starts = [0]
    seen_period = False
    for i, (t, _) in enumerate(toks):
        in_punct_chars = t in PUNCT_CHARS
        if seen_period and not is_punct(t) and not in_punct_chars:
            starts.append(i)
            seen_period = False
        elif in_punct_chars:
            seen_period = True
    out = []
    for s, i0 in enumerate(starts):
        i1 = starts[s + 1] if s + 1 < len(starts) else len(toks)
        last, last_pos = toks[i1 - 1]
        out.append((text[toks[i0][1]:last_pos + len(last)], toks[i0][1]))
    return out
