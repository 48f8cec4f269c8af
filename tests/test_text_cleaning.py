"""clean() in front of phonemisation (zonos/conditioning.py:199-288).  The reference spells numbers with `inflect`, which is not
installed here: the restatement is checked against the reference's own docstring example (conditioning.py:213-214), against
inflect's documented output format for the calls the reference makes, and - when inflect IS importable - word for word against it."""
import pytest

from zonos_amd import text_cleaning as tc
from zonos_amd._lib import ZonosHipError


def test_reference_docstring_example():
    assert tc.normalize_numbers("I have $5.50 and it's 3rd place") == "I have five dollars, fifty cents and it's third place"


@pytest.mark.parametrize("text,want", [
    ("42", "forty-two"), ("I am 42.", "I am forty-two."), ("100", "one hundred"), ("101", "one hundred one"),
    ("1,234,567 people", "one million, two hundred thirty-four thousand, five hundred sixty-seven people"),
    ("in 1984", "in nineteen eighty-four"), ("1905", "nineteen oh five"), ("1900", "nineteen hundred"), ("2000", "two thousand"),
    ("2007", "two thousand seven"), ("2010", "twenty ten"), ("3000", "three thousand"), ("1000", "one thousand"),
    ("3.14", "three point fourteen"), ("£20", "twenty pounds"), ("$1", "one dollar"), ("$0.01", "one cent"), ("$2.05", "two dollars, five cents"),
    ("1st 2nd 3rd 4th", "first second third fourth"), ("21st", "twenty-first"), ("12th", "twelfth"), ("40th", "fortieth"), ("100th", "one hundredth"),
    ("no digits here!", "no digits here!"), ("0", "zero"),
    # ordinals above 100 keep inflect's default "and" (the reference passes no andword there: conditioning.py:180-181); parity unpinned
    ("101st", "one hundred and first"), ("121st", "one hundred and twenty-first"), ("1234th", "one thousand, two hundred and thirty-fourth"),
    ("1001st", "one thousand and first"), ("2300th", "two thousand, three hundredth"),
])
def test_number_normalisation_known_answers(text, want):
    assert tc.normalize_numbers(text) == want


def test_matches_inflect_when_installed():
    inflect = pytest.importorskip("inflect")
    eng = inflect.engine()
    for n in list(range(0, 130)) + [999, 1000, 1001, 1100, 12345, 100000, 1000000, 987654321]:
        assert tc.cardinal(n) == eng.number_to_words(n, andword=""), n
        assert tc.ordinal(n) == eng.number_to_words(f"{n}th"), n
    for y in range(1001, 3000, 7):
        assert tc.digit_pairs(str(y)) == eng.number_to_words(y, andword="", zero="oh", group=2), y


def test_clean_routes_by_language_and_never_passes_japanese_through_unnormalised():
    assert tc.clean(["7 cats", "hi"], ["en-us", "de"]) == ["seven cats", "hi"]
    try:
        import kanjize, sudachipy  # noqa: F401
    except ImportError:
        with pytest.raises(ZonosHipError, match="kanjize"):
            tc.clean(["2つ"], ["ja"])


def test_phonemize_cleans_before_espeak(monkeypatch):
    """The digits never reach the backend."""
    from zonos_amd import conditioning as zc
    seen = []

    class FakeBackend:
        def phonemize(self, texts, strip=True):
            seen.extend(texts)
            return ["x"]
    monkeypatch.setattr(zc, "get_backend", lambda lang: FakeBackend())
    assert zc.phonemize(["Room 42"], ["en-us"]) == ["x"]
    assert seen == ["Room forty-two"]
