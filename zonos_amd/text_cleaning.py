"""Text cleaning in front of phonemisation — what the reference's `phonemize()` does before it calls espeak
(zonos/conditioning.py:262-288 `clean`, :199-221 `normalize_numbers`, :256-260 `normalize_jp_text`).

Non-Japanese text: numbers are spelled out in English words (the reference does this for every non-"ja" language code).
Six passes, in the reference's order (:216-221): thousands separators dropped ("1,234" -> "1234"); "£12" -> "12 pounds";
"$5.50" -> "5 dollars, 50 cents" (digits stay for the later passes); decimals "3.14" -> "3 point 14"; ordinals "3rd" ->
"third"; every remaining run of digits -> words, years between 1000 and 3000 read as pairs ("1984" -> "nineteen
eighty-four", "2005" -> "two thousand five", "1900" -> "nineteen hundred").

The reference spells numbers with the third-party `inflect` package.  When `inflect` is importable it is used, so the output is
the reference's; otherwise the speller below restates inflect's published output format for exactly the calls the reference
makes (cardinals with andword="" -> "one thousand, two hundred thirty-four"; digit pairs with zero="oh"; ordinals).  inflect is
not installed in the build container: the restatement is pinned only by the reference's own docstring example and inflect's
documented examples (tests/test_text_cleaning.py) - parity unpinned beyond those.

Japanese text needs `kanjize` and `sudachipy` exactly as the reference does; without them a ZonosHipError is raised - text
is never phonemised un-normalised.
"""
from __future__ import annotations

import re
import unicodedata

from . import _lib

_ONES = ["zero", "one", "two", "three", "four", "five", "six", "seven", "eight", "nine", "ten", "eleven", "twelve", "thirteen", "fourteen",
         "fifteen", "sixteen", "seventeen", "eighteen", "nineteen"]
_TENS = ["", "", "twenty", "thirty", "forty", "fifty", "sixty", "seventy", "eighty", "ninety"]
_SCALES = ["", "thousand", "million", "billion", "trillion", "quadrillion", "quintillion", "sextillion", "septillion", "octillion", "nonillion",
           "decillion"]
_ORDINAL_WORD = {"one": "first", "two": "second", "three": "third", "five": "fifth", "eight": "eighth", "nine": "ninth", "twelve": "twelfth"}


def _below_hundred(n: int) -> str:
    if n < 20:
        return _ONES[n]
    return _TENS[n // 10] + ("-" + _ONES[n % 10] if n % 10 else "")


def _below_thousand(n: int) -> str:
    if n < 100:
        return _below_hundred(n)
    rest = n % 100
    return _ONES[n // 100] + " hundred" + (" " + _below_hundred(rest) if rest else "")


def cardinal(n: int) -> str:
    """inflect number_to_words(n, andword=""): three-digit groups joined by ", ", no "and"."""
    if n == 0:
        return "zero"
    groups, i = [], 0
    while n:
        n, g = divmod(n, 1000)
        if g:
            if i >= len(_SCALES):
                raise _lib.ZonosHipError("number too large to spell out")
            groups.append(_below_thousand(g) + (" " + _SCALES[i] if i else ""))
        i += 1
    return ", ".join(reversed(groups))


def digit_pairs(digits: str) -> str:
    """inflect number_to_words(n, andword="", zero="oh", group=2): the digits read two at a time ("1984" -> "nineteen,
    eighty-four", "1905" -> "nineteen, oh five")."""
    out = []
    for i in range(0, len(digits), 2):
        pair = digits[i:i + 2]
        if len(pair) == 1:
            out.append("oh" if pair == "0" else _ONES[int(pair)])
        elif pair[0] == "0":
            out.append("oh " + ("oh" if pair[1] == "0" else _ONES[int(pair[1])]))
        else:
            out.append(_below_hundred(int(pair)))
    return ", ".join(out)


def cardinal_and(n: int) -> str:
    """inflect number_to_words(n) with its default andword="and": "and" between a group's hundreds and its tens / units
    ("one thousand, two hundred and thirty-four"), and in place of the last comma when the last group has no hundreds
    ("one thousand and one")."""
    if n == 0:
        return "zero"
    groups, i = [], 0
    while n:
        n, g = divmod(n, 1000)
        if g:
            if i >= len(_SCALES):
                raise _lib.ZonosHipError("number too large to spell out")
            h, rest = divmod(g, 100)
            said = (_ONES[h] + " hundred" + (" and " + _below_hundred(rest) if rest else "")) if h else _below_hundred(rest)
            groups.append((said + (" " + _SCALES[i] if i else ""), i == 0 and h == 0))
        i += 1
    groups.reverse()
    out = groups[0][0]
    for said, bare_tail in groups[1:]:
        out += (" and " if bare_tail else ", ") + said
    return out


def ordinal(n: int) -> str:
    """inflect number_to_words("<n>th") as the reference calls it (conditioning.py:180-181: default andword, so "121st" -> "one hundred
    and twenty-first", "1234th" -> "one thousand, two hundred and thirty-fourth"): the cardinal with its last word made ordinal."""
    words = cardinal_and(n)
    head, sep, last = words.rpartition(" ")
    stem, hy, tail = last.rpartition("-")
    if tail in _ORDINAL_WORD:
        tail = _ORDINAL_WORD[tail]
    elif tail.endswith("y"):
        tail = tail[:-1] + "ieth"
    else:
        tail = tail + "th"
    return head + sep + stem + hy + tail


def _speller():
    """(cardinal, digit_pairs, ordinal) from inflect when it is installed (the reference's exact words), else the restatement."""
    try:
        import inflect
    except ImportError:
        return cardinal, digit_pairs, ordinal
    eng = inflect.engine()
    return (lambda n: eng.number_to_words(n, andword=""), lambda ds: eng.number_to_words(int(ds), andword="", zero="oh", group=2),
            lambda n: eng.number_to_words(f"{n}th"))


def _money(amount: str) -> str:
    parts = amount.split(".")
    if len(parts) > 2:
        return amount + " dollars"
    dollars = int(parts[0]) if parts[0] else 0
    cents = int(parts[1]) if len(parts) > 1 and parts[1] else 0
    said = []
    if dollars:
        said.append(f"{dollars} dollar" + ("" if dollars == 1 else "s"))
    if cents:
        said.append(f"{cents} cent" + ("" if cents == 1 else "s"))
    return ", ".join(said) if said else "zero dollars"


def normalize_numbers(text: str) -> str:
    """conditioning.py:199-221.  "I have $5.50 and it's 3rd place" -> "I have five dollars, fifty cents and it's third place"."""
    card, pairs, ordn = _speller()

    def number(m: re.Match) -> str:
        n = int(m.group(0))
        if 1000 < n < 3000:
            if n == 2000:
                return "two thousand"
            if 2000 < n < 2010:
                return "two thousand " + card(n % 100)
            if n % 100 == 0:
                return card(n // 100) + " hundred"
            return pairs(m.group(0).lstrip("0") or "0").replace(", ", " ")
        return card(n)

    text = re.sub(r"([0-9][0-9\,]+[0-9])", lambda m: m.group(1).replace(",", ""), text)
    text = re.sub(r"£([0-9\,]*[0-9]+)", r"\1 pounds", text)
    text = re.sub(r"\$([0-9\.\,]*[0-9]+)", lambda m: _money(m.group(1)), text)
    text = re.sub(r"([0-9]+\.[0-9]+)", lambda m: m.group(1).replace(".", " point "), text)
    text = re.sub(r"[0-9]+(st|nd|rd|th)", lambda m: ordn(int(re.match(r"[0-9]+", m.group(0)).group(0))), text)
    return re.sub(r"[0-9]+", number, text)


_jp_tokenizer = None


def normalize_jp_text(text: str) -> str:
    """conditioning.py:256-260: NFKC, digits -> kanji numerals, sudachi reading forms joined by spaces."""
    global _jp_tokenizer
    try:
        from kanjize import number2kanji
        from sudachipy import Dictionary, SplitMode
    except ImportError as e:
        raise _lib.ZonosHipError("Japanese text needs `kanjize` and `sudachipy` (+ its full dictionary), as the reference does; "
                                 "pass espeak=('phonemes', [...]) or ('ids', tensor) instead") from e
    if _jp_tokenizer is None:
        _jp_tokenizer = Dictionary(dict="full").create()
    text = unicodedata.normalize("NFKC", text)
    text = re.sub(r"\d+", lambda m: number2kanji(int(m[0])), text)
    return " ".join(x.reading_form() for x in _jp_tokenizer.tokenize(text, SplitMode.A))


def clean(texts: list[str], languages: list[str]) -> list[str]:
    """conditioning.py:262-288."""
    return [normalize_jp_text(t) if "ja" in lang else normalize_numbers(t) for t, lang in zip(texts, languages)]
