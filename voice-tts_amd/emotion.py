"""Emotion labels -> the 8-dimensional emotion vector `IndexTTS2.infer(emo_vector=...)` takes.

Mirrors the behaviour of the reference's `emotion.py` (`create_emotion_vector`, :279-327; order of the dimensions :27;
unknown labels fall back to "calm", :206-212; a dimension named twice keeps the larger value, :238-243).  The synonym
table below is this repository's own (the eight canonical names in both languages as the `/tts` request documents them,
`server.py:189-199`, plus everyday synonyms); the reference's docstring examples (:269-273, :300-304) are the known
answers `tests/test_server.py` checks.
"""
import logging

logger = logging.getLogger("indextts.emotion")

STANDARD_EMOTION_ORDER = ["happy", "angry", "sad", "afraid", "disgusted", "melancholic", "surprised", "calm"]

_SYNONYMS = {
    "happy": ("happy happiness joy joyful cheerful delighted pleased excited glad "
              "高兴 快乐 开心 愉快 欢乐 喜悦 兴奋 欣喜"),
    "angry": ("angry anger mad furious irritated annoyed enraged rage "
              "愤怒 生气 发怒 恼怒 气愤 暴怒 恼火"),
    "sad": ("sad sadness sorrow sorrowful unhappy grief upset "
            "悲伤 难过 忧伤 伤心 悲痛 哀伤 悲哀"),
    "afraid": ("afraid fear fearful scared frightened terrified panic "
               "恐惧 害怕 恐慌 惊恐 畏惧 惧怕"),
    "disgusted": ("disgusted disgust disgusting revolted repulsed "
                  "反感 厌恶 恶心 嫌弃 讨厌 憎恶"),
    "melancholic": ("melancholic melancholy depressed gloomy down dejected low "
                    "低落 忧郁 沮丧 消沉 郁闷 抑郁"),
    "surprised": ("surprised surprise astonished amazed shocked startled "
                  "惊讶 吃惊 震惊 惊奇 诧异 惊喜"),
    "calm": ("calm neutral peaceful relaxed natural serene "
             "平静 自然 淡定 平和 安静 宁静 放松 冷静 中性"),
}
EMOTION_MAPPING = {w: std for std, words in _SYNONYMS.items() for w in words.split()}


def normalize_emotion_label(label):
    """Any known synonym (case-insensitive, surrounding blanks ignored) -> one of the 8 standard names; unknown -> "calm"."""
    std = EMOTION_MAPPING.get(label.strip().lower())
    if std is None:
        logger.warning(f"Unknown emotion label '{label}', defaulting to 'calm'")
        return "calm"
    return std


def normalize_emotion_dict(emotion_input):
    out = {e: 0.0 for e in STANDARD_EMOTION_ORDER}
    for label, value in emotion_input.items():
        std = normalize_emotion_label(label)
        out[std] = max(out[std], float(value))
    return out


def emotion_dict_to_vector(emotion_dict):
    return [emotion_dict.get(e, 0.0) for e in STANDARD_EMOTION_ORDER]


def create_emotion_vector(emotion_input, alpha=1.0):
    """str (one label, strength `alpha`) or dict label -> strength  =>  list of 8 floats in STANDARD_EMOTION_ORDER."""
    if isinstance(emotion_input, str):
        return emotion_dict_to_vector(normalize_emotion_dict({normalize_emotion_label(emotion_input): alpha}))
    if isinstance(emotion_input, dict):
        return emotion_dict_to_vector(normalize_emotion_dict(emotion_input))
    raise TypeError(f"emotion_input must be str or dict, got {type(emotion_input)}")
