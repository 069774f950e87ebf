"""Shim package: `from indextts.infer_v2 import IndexTTS2` resolves to the MI355X-native mirror
(`voice-tts_amd/infer_v2.py`), so `server.py` of the reference imports it unchanged (server.py:52-68)."""
