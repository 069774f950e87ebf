"""Emotion labels -> the 8-dimensional emotion vector `IndexTTS2.infer(emo_vector=...)` takes.

Mirrors the reference's `emotion.py`: `create_emotion_vector` (:257-290), the order of the dimensions (:27), unknown
labels fall back to "calm" with a warning (:206-212), a dimension named twice keeps the larger value (:238-243).  The
label vocabulary is part of the `/tts` wire contract (which of the 8 dimensions a request's `emotion` string drives), so the
table below holds exactly the reference's 119 labels (`EMOTION_MAPPING`, :31-187) -- no more, no fewer;
`tests/golden/emotion_labels.json` (written from the imported reference module by tests/golden/make_golden.py) freezes every
label and `tests/test_server.py` checks all of them.
"""
import logging

logger = logging.getLogger("indextts.emotion")

STANDARD_EMOTION_ORDER = ["happy", "angry", "sad", "afraid", "disgusted", "melancholic", "surprised", "calm"]

_LABELS = {
    "happy": "happy happiness joy joyful cheerful delighted pleased excited 高兴 快乐 开心 愉快 欢乐 喜悦 兴奋 欣喜",
    "angry": "angry anger mad furious irritated annoyed enraged 愤怒 生气 发怒 恼怒 气愤 火大",
    "sad": "sad sadness unhappy sorrow sorrowful grief heartbroken 悲伤 难过 伤心 忧伤 哀伤 痛苦 悲痛",
    "afraid": "afraid fear fearful scared frightened terrified anxious nervous panic panicked 恐惧 害怕 恐慌 惊恐 畏惧 紧张",
    "disgusted": "disgusted disgust disgusting repulsed revolted nauseated 反感 厌恶 恶心 讨厌 反胃 嫌弃",
    "melancholic": "melancholic melancholy depressed depression gloomy downcast dejected despondent 低落 忧郁 沮丧 消沉 抑郁 颓废 低沉",
    "surprised": "surprised surprise astonished amazed shocked startled stunned 惊讶 吃惊 震惊 惊奇 诧异 惊诧 愕然",
    "calm": "calm normal calmness peaceful serene tranquil relaxed composed neutral natural 平静 自然 淡定 平和 安静 宁静 放松 冷静 中性",
}
EMOTION_MAPPING = {w: std for std, words in _LABELS.items() for w in words.split()}


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
