"""Importable alias for the `voice-tts_amd/` package directory.

The product package lives in `voice-tts_amd/` (the layout name cannot be a Python
identifier); this shim makes it importable as `voice_tts_amd` by pointing the package
search path at that directory.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "voice-tts_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
