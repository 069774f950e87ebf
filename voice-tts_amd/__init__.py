"""MI355X-native IndexTTS2 hot path (GPT decode + BigVGAN) behind the reference's seams.

Layout:
  csrc/          hand-written HIP kernels for gfx950 + the C-ABI (include/ixtts_hip.h)
  _lib.py        ctypes binding of libixtts_hip.so (fails loudly when the library is missing)
  weights.py     state-dict recipes / loaders / weight-norm fold
  aa_activation.py, bigvgan.py   vocoder seam  (reference: indextts/s2mel/modules/bigvgan)
  gpt_engine.py  decode-engine seam (reference: indextts/gpt/model_v2.py GPT2InferenceModel)
"""
__version__ = "0.1.0"
