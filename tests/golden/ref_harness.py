"""Import harness for the read-only reference at /root/reference (golden generation ONLY).

TEST INFRASTRUCTURE.  Used exclusively by tests/golden/make_golden.py, in the build
container, to instantiate the reference's own module classes with seeded random weights
and dump small input/output fixtures.  Nothing here ships to the GPU box as a dependency:
tests, bench.py and the product path read only the committed .npz fixtures.

The reference pins transformers==4.52.1 (pyproject.toml:58); this image has 5.x, so a
handful of names the vendored files import at module load are missing.  We inject inert
placeholders for them.  None of the placeholders is ever *called*
on the paths we exercise (GPT2InferenceModel.forward, BigVGAN.forward, Activation1d).
"""
import importlib
import sys
import types

REF = "/root/reference"


def _mod(name, **attrs):
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        sys.modules[name] = m
    for k, v in attrs.items():
        if not hasattr(m, k):
            setattr(m, k, v)
    return m


def _ensure(modname, **attrs):
    try:
        m = importlib.import_module(modname)
    except Exception:
        m = _mod(modname)
        parent, _, child = modname.rpartition(".")
        if parent and parent in sys.modules:
            setattr(sys.modules[parent], child, m)
    for k, v in attrs.items():
        if not hasattr(m, k):
            setattr(m, k, v)
    return m


class _Placeholder:
    def __init__(self, *a, **k):
        raise RuntimeError("placeholder for a symbol absent from this transformers version")


def install():
    """Put /root/reference on sys.path and install the import-time placeholders."""
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)

    import transformers  # noqa: F401  (before the stubs: its lazy loader probes package specs)

    # --- absent third-party packages: bare stubs (never called on our paths) ---------
    if "torchaudio" not in sys.modules:
        try:
            import torchaudio  # noqa: F401
        except Exception:
            ta = _mod("torchaudio", load=None)
            ta.functional = _mod("torchaudio.functional")
            ta.functional.__path__ = []
            # kmeans/vocos.py:14 imports two mel helpers for feature classes the codec path never builds
            ta.functional.functional = _mod("torchaudio.functional.functional", _hz_to_mel=None, _mel_to_hz=None)
            ta.transforms = _mod("torchaudio.transforms")
            ta.compliance = _mod("torchaudio.compliance")
            ta.compliance.kaldi = _mod("torchaudio.compliance.kaldi")
    try:
        import munch  # noqa: F401
    except Exception:
        _mod("munch", Munch=dict)
    try:
        import librosa  # noqa: F401
    except Exception:
        lb = _mod("librosa")
        lb.util = _mod("librosa.util", normalize=None)
        lb.filters = _mod("librosa.filters", mel=None)
    # skip `import audiotools` in indextts/s2mel/dac/__init__.py:6
    for pk in ("indextts.s2mel.dac", "indextts.s2mel.dac.nn"):
        if pk not in sys.modules:
            m = _mod(pk)
            m.__path__ = [REF + "/" + pk.replace(".", "/")]

    # --- names removed in transformers 5.x that the vendored files import ------------
    _ensure("transformers.cache_utils", OffloadedCache=_Placeholder, QuantizedCacheConfig=_Placeholder)
    _ensure(
        "transformers.pytorch_utils",
        isin_mps_friendly=lambda e, t: __import__("torch").isin(e, t),
        find_pruneable_heads_and_indices=_Placeholder,
        prune_conv1d_layer=_Placeholder,
        prune_layer=_Placeholder,
    )
    _ensure("transformers.tokenization_utils", ExtensionsTrie=_Placeholder)
    _ensure(
        "transformers.generation.beam_constraints",
        DisjunctiveConstraint=_Placeholder,
        PhrasalConstraint=_Placeholder,
        Constraint=_Placeholder,
        ConstraintListState=_Placeholder,
    )
    _ensure("transformers.generation.candidate_generator", _crop_past_key_values=_Placeholder)
    _ensure(
        "transformers.generation.configuration_utils",
        NEED_SETUP_CACHE_CLASSES_MAPPING={},
        QUANT_BACKEND_CLASSES_MAPPING={},
    )
    _ensure("transformers.generation.logits_process", HammingDiversityLogitsProcessor=_Placeholder)
    _ensure(
        "transformers.utils",
        FLAX_WEIGHTS_NAME="flax_model.msgpack",
        TF2_WEIGHTS_NAME="tf_model.h5",
        TF_WEIGHTS_NAME="model.ckpt",
        download_url=lambda *a, **k: None,
        is_offline_mode=lambda: True,
        is_remote_url=lambda *a, **k: False,
        is_safetensors_available=lambda: True,
        is_torch_sdpa_available=lambda: True,
    )
    _ensure("transformers.modeling_tf_pytorch_utils")
    _ensure("transformers.modeling_flax_pytorch_utils")
    _ensure(
        "transformers.utils.model_parallel_utils",
        assert_device_map=_Placeholder,
        get_device_map=_Placeholder,
    )
    _ensure("transformers.modeling_utils", SequenceSummary=_Placeholder)
    # vendored scorer text stands in for the removed third-party module
    if "transformers.generation.beam_search" not in sys.modules:
        try:
            importlib.import_module("transformers.generation.beam_search")
        except Exception:
            bs = importlib.import_module("indextts.gpt.transformers_beam_search")
            sys.modules["transformers.generation.beam_search"] = bs
