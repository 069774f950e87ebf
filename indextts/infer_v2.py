"""Drop-in module path of the reference (`indextts/infer_v2.py`): re-exports the HIP-backed mirror."""
from voice_tts_amd.infer_v2 import Glue, IndexTTS2  # noqa: F401
