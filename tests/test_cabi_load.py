"""CPU-side: the C-ABI library loads and exports every symbol include/ixtts_hip.h declares."""
import os
import re

import pytest

from voice_tts_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "ixtts_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(ixtts_[a-z0-9_]+)\s*\(", text))


def test_header_and_binding_agree():
    assert _declared() == set(_lib.SYMBOLS)


def test_library_exports_every_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        from voice_tts_amd import build
        build.build(verbose=False)
    L = _lib.lib()
    for name in _declared():
        assert hasattr(L, name), name
    assert b"gfx950" in L.ixtts_version()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.IxttsError):
        _lib.lib()
