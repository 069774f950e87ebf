"""Text front-end: normalisation rules, sentencepiece tokenisation, segment splitting.

Host-side mirror of `indextts/utils/front.py` (`TextNormalizer` :11-229, `TextTokenizer` :231-436) and of the two CJK
helpers of `indextts/utils/common.py` (:29-82), with the reference's names, arguments and results, so that
`IndexTTS2.infer` can take raw text exactly as the reference does (`infer_v2.py:161-165,582-617`).

* The number / date / unit verbaliser is third-party (WeTextProcessing `tn` on Linux, `wetext` elsewhere,
  `front.py:88-111`) and is not in this image: `TextNormalizer.load()` imports it exactly as the reference does and
  raises `ImportError` when it is missing; `TextNormalizer(zh_normalizer=..., en_normalizer=...)` injects any object with a
  `normalize(str) -> str` method instead.  Everything the reference itself adds around that verbaliser — language choice,
  `'s` expansion, protection of pinyin-with-tone and of dotted personal names, punctuation folding — is implemented here
  and pinned to fixtures produced by the reference's class with an identity verbaliser on both sides
  (`tests/golden/front.json`); the verbaliser's own output is not covered ("parity unpinned" for digits-to-words).
* `split_segments_by_token` reproduces the reference's segmentation decisions including its corner behaviour (a quote
  after a sentence end is kept with the sentence AND opens the next one, `front.py:378-382`; segments merge while they
  fit, or fit half the limit once `quick_streaming_tokens` has passed).
"""
import os
import re
import traceback
import warnings

_CJK = re.compile(r"([\u1100-\u11ff\u2e80-\ua4cf\ua840-\ud7af\uf900-\ufaff\ufe30-\ufe4f\uff65-\uffdc\U00020000-\U0002FFFF])")


def tokenize_by_CJK_char(line, do_upper_case=True):
    """"你好世界是 hello world 的中文" -> "你 好 世 界 是 HELLO WORLD 的 中 文" (common.py:29-51)."""
    parts = (p.strip() for p in _CJK.split(line.strip()))
    return " ".join(p.upper() if do_upper_case else p for p in parts if p)


_LATIN_RUN = re.compile(r"([A-Z]+(?:[\s-][A-Z-]+)*)", re.IGNORECASE)
_SENT_MARK = re.compile(r"^.*?(<sent_(\d+)>)")


def de_tokenized_by_CJK_char(line, do_lower_case=False):
    """"你 好 世 界 是 HELLO WORLD 的 中 文" -> "你好世界是 hello world 的中文" (common.py:54-82).

    Runs of Latin words are shielded behind `<sent_i>` marks while the blanks between CJK characters are dropped; a word
    gets back the run of the FIRST mark it contains (the reference's behaviour, kept).
    """
    runs = _LATIN_RUN.findall(line)
    for i, run in enumerate(runs):
        line = line.replace(run, f"<sent_{i}>")
    words = line.split()
    for i, w in enumerate(words):
        m = _SENT_MARK.match(w)
        if m:
            w = w.replace(m.group(1), runs[int(m.group(2))])
            words[i] = w.lower() if do_lower_case else w
    return "".join(words)


def _first_seen(items):
    return list(dict.fromkeys(items))


class TextNormalizer:
    """`front.py:11-229`.  `zh_normalizer` / `en_normalizer`: objects with `normalize(str) -> str` (default: WeText, see `load`)."""

    PINYIN_TONE_PATTERN = (r"(?<![a-z])((?:[bpmfdtnlgkhjqxzcsryw]|[zcs]h)?(?:[aeiouüv]|[ae]i|u[aio]|ao|ou|i[aue]|[uüv]e"
                           r"|[uvü]ang?|uai|[aeiuv]n|[aeio]ng|ia[no]|i[ao]ng)|ng|er)([1-5])")
    NAME_PATTERN = r"[\u4e00-\u9fff]+(?:[-·—][\u4e00-\u9fff]+){1,2}"
    ENGLISH_CONTRACTION_PATTERN = r"(what|where|who|which|how|t?here|it|s?he|that|this)'s"

    def __init__(self, zh_normalizer=None, en_normalizer=None):
        self.zh_normalizer, self.en_normalizer = zh_normalizer, en_normalizer
        # punctuation folding, in the reference's order (:15-51): the order decides ties in the alternation built from the keys
        # ("，" stands before "，，，", so three full-width commas become three commas, not an ellipsis)
        groups = [("：；;，", ","), ("。", "."), ("！", "!"), ("？", "?"), ("\n", " "), ("·", "-"), ("、", ","),
                  (("...", ",,,", "，，，", "……"), "…"), ("“”\"‘’（）()《》【】[]", "'"), ("—～~", "-"), ("「」", "'"), (":", ",")]
        self.char_rep_map = {k: v for keys, v in groups for k in keys}
        self.zh_char_rep_map = {"$": ".", **self.char_rep_map}
        # alternation in insertion order, as the reference builds it on every call (:134,:143): earlier keys win a tie
        self._fold = {False: self._folder(self.char_rep_map), True: self._folder(self.zh_char_rep_map)}

    @staticmethod
    def _folder(table):
        pat = re.compile("|".join(re.escape(k) for k in table))
        return lambda s: pat.sub(lambda m: table[m.group()], s)

    def match_email(self, email):
        return re.match(r"^[a-zA-Z0-9]+@[a-zA-Z0-9]+\.[a-zA-Z]+$", email) is not None

    def use_chinese(self, s):
        """Chinese rules unless the text is Latin-script without any pinyin-with-tone (:78-86)."""
        if re.search(r"[\u4e00-\u9fff]", s) or not re.search(r"[a-zA-Z]", s) or self.match_email(s):
            return True
        return re.search(self.PINYIN_TONE_PATTERN, s, re.IGNORECASE) is not None

    def load(self):
        """Import the third-party verbalisers the way the reference does (:88-111); nothing is downloaded or faked."""
        if self.zh_normalizer is not None and self.en_normalizer is not None:
            return
        import platform

        if platform.system() != "Linux":
            from wetext import Normalizer

            self.zh_normalizer = Normalizer(remove_erhua=False, lang="zh", operator="tn")
            self.en_normalizer = Normalizer(lang="en", operator="tn")
        else:
            from tn.chinese.normalizer import Normalizer as NormalizerZh
            from tn.english.normalizer import Normalizer as NormalizerEn

            cache_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tagger_cache")
            os.makedirs(cache_dir, exist_ok=True)
            self.zh_normalizer = NormalizerZh(cache_dir=cache_dir, remove_interjections=False, remove_erhua=False, overwrite_cache=False)
            self.en_normalizer = NormalizerEn(overwrite_cache=False)

    def normalize(self, text):
        if not self.zh_normalizer or not self.en_normalizer:
            print("Error, text normalizer is not initialized !!!")
            return ""
        text_is = re.sub(self.ENGLISH_CONTRACTION_PATTERN, r"\1 is", text, flags=re.IGNORECASE)
        if self.use_chinese(text):
            shielded, pinyins = self.save_pinyin_tones(text_is.rstrip())
            shielded, names = self.save_names(shielded)
            try:
                out = self.zh_normalizer.normalize(shielded)
            except Exception:
                out = ""
                print(traceback.format_exc())
            out = self.restore_pinyin_tones(self.restore_names(out, names), pinyins)
            return self._fold[True](out)
        try:
            out = self.en_normalizer.normalize(text_is)
        except Exception:
            out = text_is
            print(traceback.format_exc())
        return self._fold[False](out)

    def correct_pinyin(self, pinyin):
        """j/q/x + u|ü -> v (ju4 -> JV4, xün1 -> XVN1); only these initials are upper-cased here (:144-155)."""
        if pinyin[0] not in "jqxJQX":
            return pinyin
        return re.sub(r"([jqx])[uü](n|e|an)*(\d)", r"\g<1>v\g<2>\g<3>", pinyin, flags=re.IGNORECASE).upper()

    # The reference numbers its placeholders in `set()` order, which varies from process to process; the result after
    # restoring is the same for any order as long as no protected string contains another, and first-seen order is used here.
    def _shield(self, pattern, text, tag):
        found = ["".join(g) if isinstance(g, tuple) else g for g in re.findall(pattern, text, re.IGNORECASE)]
        if not found:
            return text, None
        found = _first_seen(found)
        for i, s in enumerate(found):
            text = text.replace(s, f"<{tag}_{chr(ord('a') + i)}>")
        return text, found

    def save_names(self, original_text):
        return self._shield(self.NAME_PATTERN, original_text, "n")

    def restore_names(self, normalized_text, original_name_list):
        for i, name in enumerate(original_name_list or ()):
            normalized_text = normalized_text.replace(f"<n_{chr(ord('a') + i)}>", name)
        return normalized_text

    def save_pinyin_tones(self, original_text):
        return self._shield(self.PINYIN_TONE_PATTERN, original_text, "pinyin")

    def restore_pinyin_tones(self, normalized_text, original_pinyin_list):
        for i, p in enumerate(original_pinyin_list or ()):
            normalized_text = normalized_text.replace(f"<pinyin_{chr(ord('a') + i)}>", self.correct_pinyin(p))
        return normalized_text


class TextTokenizer:
    """`front.py:231-436`: sentencepiece over normalised, CJK-spaced, upper-cased text + the segment splitter."""

    punctuation_marks_tokens = [".", "!", "?", "▁.", "▁?", "▁..."]

    def __init__(self, vocab_file, normalizer=None):
        if vocab_file is None:
            raise ValueError("vocab_file is None")
        if not os.path.exists(vocab_file):
            raise ValueError(f"vocab_file {vocab_file} does not exist")
        from sentencepiece import SentencePieceProcessor

        self.vocab_file, self.normalizer = vocab_file, normalizer
        if normalizer:
            normalizer.load()
        self.sp_model = SentencePieceProcessor(model_file=vocab_file)
        self.pre_tokenizers = [tokenize_by_CJK_char]

    vocab_size = property(lambda self: self.sp_model.GetPieceSize())
    unk_token = property(lambda self: "<unk>")
    pad_token = property(lambda self: None)
    bos_token = property(lambda self: "<s>")
    eos_token = property(lambda self: "</s>")
    pad_token_id = property(lambda self: -1)
    bos_token_id = property(lambda self: 0)
    eos_token_id = property(lambda self: 1)
    unk_token_id = property(lambda self: self.sp_model.unk_id())

    @property
    def special_tokens_map(self):
        return {"unk_token": self.unk_token, "pad_token": self.pad_token, "bos_token": self.bos_token, "eos_token": self.eos_token}

    def get_vocab(self):
        return {self.convert_ids_to_tokens(i): i for i in range(self.vocab_size)}

    def convert_ids_to_tokens(self, ids):
        return self.sp_model.IdToPiece(ids)

    def convert_tokens_to_ids(self, tokens):
        return [self.sp_model.PieceToId(t) for t in ([tokens] if isinstance(tokens, str) else tokens)]

    def tokenize(self, text):
        return self.encode(text, out_type=str)

    def _prepare(self, text):
        if self.normalizer:
            text = self.normalizer.normalize(text)
        for pre in self.pre_tokenizers:
            text = pre(text)
        return text

    def encode(self, text, **kwargs):
        out_type = kwargs.pop("out_type", int)
        if len(text) == 0:
            return []
        if len(text.strip()) != 1:  # a single character goes to the model as it is (:319-320)
            text = self._prepare(text)
        return self.sp_model.Encode(text, out_type=out_type, **kwargs)

    def batch_encode(self, texts, **kwargs):
        return self.sp_model.Encode([self._prepare(t) for t in texts], out_type=kwargs.pop("out_type", int), **kwargs)

    def decode(self, ids, do_lower_case=False, **kwargs):
        ids = [ids] if isinstance(ids, int) else ids
        return de_tokenized_by_CJK_char(self.sp_model.Decode(ids, out_type=kwargs.pop("out_type", str), **kwargs), do_lower_case=do_lower_case)

    @staticmethod
    def split_segments_by_token(tokenized_str, split_tokens, max_text_tokens_per_segment, quick_streaming_tokens=0):
        """Cut a token list after each token of `split_tokens`, keep pieces within the limit, merge short neighbours (:345-421).

        A piece that would exceed the limit is re-cut at commas, then at hyphens, then by length (with a RuntimeWarning).
        """
        limit = max_text_tokens_per_segment
        commas, hyphen = [",", "▁,"], ["-"]
        may_cut_commas = not any(c in split_tokens for c in commas)
        may_cut_hyphen = "-" not in split_tokens
        pieces, cur = [], []
        n = len(tokenized_str)
        for i, tok in enumerate(tokenized_str):
            cur.append(tok)
            if may_cut_commas and any(c in cur for c in commas):
                sub = TextTokenizer.split_segments_by_token(cur, commas, limit, quick_streaming_tokens)
            elif may_cut_hyphen and "-" in cur:
                sub = TextTokenizer.split_segments_by_token(cur, hyphen, limit, quick_streaming_tokens)
            elif len(cur) <= limit:
                if tok in split_tokens and len(cur) > 2:
                    if i + 1 < n and tokenized_str[i + 1] in ("'", "▁'"):
                        cur.append(tokenized_str[i + 1])  # the closing quote stays here; it is also seen again next turn
                    pieces.append(cur)
                    cur = []
                continue
            else:
                sub = [cur[j:j + limit] for j in range(0, len(cur), limit)]
                warnings.warn(f"The tokens length of segment exceeds limit: {limit}, Tokens in segment: {cur}.Maybe unexpected behavior",
                              RuntimeWarning)
            pieces.extend(sub)
            cur = []
        if cur:
            assert len(cur) <= limit
            pieces.append(cur)
        merged, seen = [], 0
        for p in pieces:
            seen += len(p)
            if not p:
                continue
            if merged and ((len(merged[-1]) + len(p) <= limit and seen > quick_streaming_tokens) or len(merged[-1]) + len(p) <= limit / 2):
                merged[-1] = merged[-1] + p
            else:
                merged.append(p)
        return merged

    def split_segments(self, tokenized, max_text_tokens_per_segment=120, quick_streaming_tokens=0):
        return TextTokenizer.split_segments_by_token(tokenized, self.punctuation_marks_tokens, max_text_tokens_per_segment,
                                                     quick_streaming_tokens=quick_streaming_tokens)
