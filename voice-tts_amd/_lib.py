"""ctypes binding of libixtts_hip.so (the C ABI in include/ixtts_hip.h).

There is NO fallback: if the library is missing or fails to load, `lib()` raises.  The
product path never routes through oracle/ or any CPU implementation.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IXTTS_LIB") or os.path.join(HERE, "libixtts_hip.so")  # IXTTS_LIB: developer builds (trace)

MAX_STAGES = 8
MAX_RESK = 4


class BigVGANCfg(C.Structure):
    _fields_ = [
        ("num_mels", C.c_int), ("upsample_initial_channel", C.c_int), ("n_stages", C.c_int),
        ("upsample_rates", C.c_int * MAX_STAGES), ("upsample_kernel_sizes", C.c_int * MAX_STAGES),
        ("n_resblock_kernels", C.c_int), ("resblock_kernel_sizes", C.c_int * MAX_RESK),
        ("resblock_dilations", (C.c_int * 3) * MAX_RESK), ("max_frames", C.c_int), ("fast_sin", C.c_int),
    ]


class GptCfg(C.Structure):
    _fields_ = [
        ("model_dim", C.c_int), ("layers", C.c_int), ("heads", C.c_int), ("n_mel_codes", C.c_int),
        ("n_mel_pos", C.c_int), ("n_text_tokens", C.c_int), ("n_text_pos", C.c_int),
        ("start_mel_token", C.c_int), ("stop_mel_token", C.c_int), ("max_seq", C.c_int),
        ("max_batch", C.c_int), ("weight_dtype", C.c_int),
    ]


class SamplerCfg(C.Structure):
    _fields_ = [
        ("repetition_penalty", C.c_float), ("temperature", C.c_float), ("top_k", C.c_int),
        ("top_p", C.c_float), ("do_sample", C.c_int), ("suppress_stop", C.c_int), ("seed", C.c_uint64),
        ("typical_mass", C.c_float), ("length_penalty", C.c_float),
    ]


# every symbol include/ixtts_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "ixtts_version": (C.c_char_p, []),
    "ixtts_last_error": (C.c_char_p, []),
    "ixtts_aa_snake_f32": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "ixtts_attn_full_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "ixtts_attn_full_f32": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long, C.c_long, C.c_long, C.c_long,
                                      C.c_long, C.c_float, _P, C.c_size_t, _P]),
    "ixtts_adaln_rmsnorm_f32": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float, _P]),
    "ixtts_ln_modulate_f32": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float, _P]),
    "ixtts_rope_qk_f32": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "ixtts_swiglu_f32": (C.c_int, [_P, _P, C.c_long, C.c_int, _P]),
    "ixtts_wn_gate_f32": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_long, C.c_int, _P]),
    "ixtts_wn_gate_rows_f32": (C.c_int, [_P, _P, _P, C.c_long, C.c_int, C.c_long, C.c_int, C.c_long, C.c_int, _P]),
    "ixtts_reflect_halo_rows_f32": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "ixtts_gemm_x6_packed_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "ixtts_gemm_x6_pack": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "ixtts_gemm_x6_rows_padded": (C.c_long, [C.c_long]),
    "ixtts_gemm_x6_split": (C.c_int, [_P, C.c_long, _P, C.c_long, C.c_int, _P]),
    "ixtts_gemm_x6_f32": (C.c_int, [_P, C.c_long, C.c_long, _P, _P, _P, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "ixtts_bigvgan_create": (C.c_int, [C.POINTER(_P), C.POINTER(BigVGANCfg)]),
    "ixtts_bigvgan_set_tensor": (C.c_int, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int]),
    "ixtts_bigvgan_finalize": (C.c_int, [_P]),
    "ixtts_bigvgan_arena": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "ixtts_bigvgan_adopt_arena": (C.c_int, [_P]),
    "ixtts_bigvgan_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P]),
    "ixtts_bigvgan_flops": (C.c_double, [_P, C.c_int, C.c_int]),
    "ixtts_bigvgan_destroy": (C.c_int, [_P]),
    "ixtts_gpt_create": (C.c_int, [C.POINTER(_P), C.POINTER(GptCfg)]),
    "ixtts_gpt_set_tensor": (C.c_int, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int]),
    "ixtts_gpt_finalize": (C.c_int, [_P]),
    "ixtts_gpt_arena": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "ixtts_gpt_adopt_arena": (C.c_int, [_P]),
    "ixtts_gpt_share_arena": (C.c_int, [_P, _P]),
    "ixtts_gpt_prefill": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, _P]),
    "ixtts_gpt_decode": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(SamplerCfg), _P]),
    "ixtts_gpt_read": (C.c_int, [_P, C.c_int, _P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), _P]),
    "ixtts_gpt_read_logits": (C.c_int, [_P, C.c_int, _P, _P]),
    "ixtts_gpt_beam_begin": (C.c_int, [_P, C.c_int, _P]),
    "ixtts_gpt_beam_decode": (C.c_int, [_P, C.c_int, C.POINTER(SamplerCfg), _P]),
    "ixtts_gpt_beam_read": (C.c_int, [_P, C.c_int, _P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float), _P, _P, _P, _P]),
    "ixtts_gpt_beam_force": (C.c_int, [_P, _P, C.c_int, _P]),
    "ixtts_gpt_beam_begin_group": (C.c_int, [_P, C.c_int, C.c_int, C.c_uint64, _P]),
    "ixtts_gpt_beam_park_group": (C.c_int, [_P, C.c_int, _P]),
    "ixtts_gpt_beam_decode_groups": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(SamplerCfg), _P]),
    "ixtts_gpt_beam_read_group": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float), _P, _P, _P, _P]),
    "ixtts_gpt_beam_force_group": (C.c_int, [_P, C.c_int, _P, C.c_int, _P]),
    "ixtts_gpt_read_probs": (C.c_int, [_P, C.c_int, _P, _P]),
    "ixtts_gpt_force_next": (C.c_int, [_P, C.c_int, C.c_int32, _P]),
    "ixtts_gpt_latent": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, _P, _P]),
    "ixtts_gpt_bench_gemv": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P]),
    "ixtts_gpt_step_bytes": (C.c_double, [_P, C.c_int, C.c_int]),
    "ixtts_gpt_max_batch": (C.c_int, []),
    "ixtts_gpt_destroy": (C.c_int, [_P]),
}

_lib = None


class IxttsError(RuntimeError):
    pass


def lib():
    """Load libixtts_hip.so and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its bundled HIP runtime must be the one this process initialises.  Loading libixtts_hip.so before torch
    # pulls in /opt/rocm's libamdhip64 as a second runtime, and device memory calls through it then fail with
    # "no ROCm-capable device is detected" (seen when build() ran before smoke() in one process).
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise IxttsError(
            f"{LIB_PATH} not found: build it with `python -m voice_tts_amd.build` "
            "(there is no CPU fallback for the HIP path)")
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc, what=""):
    if rc != 0:
        msg = lib().ixtts_last_error().decode(errors="replace")
        raise IxttsError(f"{what} failed with code {rc}: {msg}")


def max_batch():
    """Decode slots the engine can step together (a build constant of libixtts_hip.so)."""
    return int(lib().ixtts_gpt_max_batch())


def current_stream_ptr():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
