"""Text front-end (N4) against fixtures made by the reference's own TextNormalizer / TextTokenizer / CJK helpers
(tests/golden/make_golden.py gen_front; the absent WeText verbalisers replaced by identity on both sides)."""
import json
import os
import warnings

import pytest

from voice_tts_amd import front as F

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def G():
    with open(os.path.join(HERE, "golden", "front.json"), encoding="utf-8") as f:
        return json.load(f)


class Identity:
    def normalize(self, text):
        return text


@pytest.fixture(scope="module")
def norm():
    return F.TextNormalizer(zh_normalizer=Identity(), en_normalizer=Identity())


def test_docstring_known_answers():
    # common.py:35-37, 56-62; front.py:146-147
    assert F.tokenize_by_CJK_char("你好世界是 hello world 的中文") == "你 好 世 界 是 HELLO WORLD 的 中 文"
    # (the docstring of de_tokenized_by_CJK_char shows blanks around the Latin run that its code does not produce; the
    # fixture holds what the code returns)
    assert F.de_tokenized_by_CJK_char("SEE YOU!", do_lower_case=True) == "see you!"
    n = F.TextNormalizer()
    assert [n.correct_pinyin(p) for p in ("ju4", "que4", "xün1")] == ["JV4", "QVE4", "XVN1"]
    for bad in ("beta1", "better1", "voice2", "bala2", "babala2", "hunger2"):  # front.py:506-510
        import re
        assert re.match(F.TextNormalizer.PINYIN_TONE_PATTERN, bad, re.IGNORECASE) is None


def test_normalizer_rules(G, norm):
    for c in G["normalize"]:
        assert norm.use_chinese(c["text"]) == c["use_chinese"], c["text"]
        assert norm.normalize(c["text"]) == c["normalized"], c["text"]
    for c in G["pinyin"]:
        assert norm.correct_pinyin(c["pinyin"]) == c["corrected"]


def test_normalizer_without_verbaliser():
    n = F.TextNormalizer()
    assert n.normalize("我爱你！") == ""  # "not initialized" path (front.py:114-116)
    with pytest.raises(ImportError):  # WeText is not in this image; nothing stands in for it
        n.load()


def test_cjk_helpers(G):
    for c in G["cjk"]:
        assert F.tokenize_by_CJK_char(c["text"]) == c["spaced"]
        assert F.tokenize_by_CJK_char(c["text"], do_upper_case=False) == c["spaced_keep_case"]
        assert F.de_tokenized_by_CJK_char(c["spaced"]) == c["joined"]
        assert F.de_tokenized_by_CJK_char(c["spaced"], do_lower_case=True) == c["joined_lower"]


def test_tokenizer(G, norm):
    tok = F.TextTokenizer(os.path.join(HERE, "golden", "tiny_bpe.model"), norm)
    m = G["tokenizer_meta"]
    assert tok.vocab_size == m["vocab_size"] and tok.unk_token_id == m["unk_token_id"] and tok.special_tokens_map == m["special_tokens_map"]
    assert tok.convert_tokens_to_ids(list(F.TextTokenizer.punctuation_marks_tokens)) == m["punct_ids"]
    assert (tok.bos_token_id, tok.eos_token_id, tok.pad_token_id) == (0, 1, -1)
    for c in G["tokenizer"]:
        assert tok.encode(c["text"]) == c["ids"], c["text"]
        toks = tok.tokenize(c["text"])
        assert toks == c["tokens"]
        assert tok.convert_tokens_to_ids(toks) == c["ids"]
        if c["ids"]:
            assert tok.decode(c["ids"]) == c["decoded"]
            assert tok.decode(c["ids"], do_lower_case=True) == c["decoded_lower"]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert tok.split_segments(toks, 20) == c["segments_20"]
            assert tok.split_segments(toks) == c["segments_120"]
    texts = [c["text"] for c in G["normalize"] if c["text"]][:6]
    assert tok.batch_encode(texts) == G["tokenizer_batch"]
    with pytest.raises(ValueError):
        F.TextTokenizer(None)
    with pytest.raises(ValueError):
        F.TextTokenizer("/nonexistent/bpe.model")


def test_split_segments_random_streams(G):
    marks = list(F.TextTokenizer.punctuation_marks_tokens)
    for c in G["split"]:
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            got = F.TextTokenizer.split_segments_by_token(c["tokens"], marks, c["limit"], c["quick"])
        assert got == c["segments"], (c["limit"], c["quick"], c["tokens"])
        assert any(issubclass(w.category, RuntimeWarning) for w in caught) == c["warned"]
