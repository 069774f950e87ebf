"""Prompt-side glue (voice-tts_amd/prompt.py; infer_v2.py:307-419,508-580) against tests/golden/prompt_tiny.npz (the reference's own
CAMPPlus, RepCodec.quantize and mel_spectrogram, make_golden.py `gen_prompt`) and against independent implementations where
the reference only calls an absent library (DESIGN.md section 2 lists what stays "parity unpinned")."""
import io
import math
import struct
import wave

import numpy as np
import pytest
import torch

import voice_tts_amd.prompt as PR


def test_camplus_vs_reference(golden):
    g = golden("prompt_tiny.npz")
    cam = PR.CamPlus(PR.make_camplus_weights(seed=int(g["seeds"][0])))
    for T in (215, 57):
        got = cam(torch.from_numpy(g[f"cam_feat_{T}"]))
        ref = torch.from_numpy(g[f"cam_style_{T}"])
        assert got.shape == ref.shape == (1, 192)
        assert (got - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


def test_semantic_codec_quantize_vs_reference(golden):
    g = golden("prompt_tiny.npz")
    keys = ["codebook_size", "hidden_size", "codebook_dim", "vocos_dim", "vocos_intermediate_dim", "vocos_num_layers"]
    cfg = dict(zip(keys, (int(v) for v in g["codec_cfg"])))
    codec = PR.SemanticCodec(PR.make_codec_weights(cfg, seed=int(g["seeds"][1])), cfg)
    codes, S = codec.quantize(torch.from_numpy(g["codec_x"]))
    assert torch.equal(codes, torch.from_numpy(g["codec_codes"]))  # index work: exact
    assert (S - torch.from_numpy(g["codec_S"])).abs().max().item() <= 1e-5
    # a checkpoint that still carries weight_g / weight_v (as model.safetensors does) folds to the same tensors
    W = PR.make_codec_weights(cfg, seed=int(g["seeds"][1]))
    for n in ("in_project", "out_project"):
        w = W.pop(f"quantizer.quantizers.0.{n}.weight")
        W[f"quantizer.quantizers.0.{n}.weight_g"] = w.reshape(w.shape[0], -1).norm(dim=1).reshape(-1, 1, 1)
        W[f"quantizer.quantizers.0.{n}.weight_v"] = 2.5 * w
    c2, S2 = PR.SemanticCodec(W, cfg).quantize(torch.from_numpy(g["codec_x"]))
    assert torch.equal(c2, codes) and torch.allclose(S2, S, atol=1e-6)
    assert set(codec.s2mel_quantizer_tensors()) == {"quantizer.codebook.weight", "quantizer.out_project.weight", "quantizer.out_project.bias"}


def test_mel_spectrogram_vs_reference_function(golden):
    g = golden("prompt_tiny.npz")
    got = PR.mel_spectrogram(torch.from_numpy(g["mel_y"]))
    ref = torch.from_numpy(g["mel_out"])
    assert got.shape == ref.shape == (1, 80, 22050 // 256)
    assert (got - ref).abs().max().item() <= 1e-5


def test_slaney_mel_basis_published_properties():
    """librosa is absent (parity unpinned by library output): hold the basis to the published definition instead --
    unit-area triangles in Hz on the Slaney scale, linear below 1 kHz, centres monotone, full band covered."""
    sr, n_fft, n = 22050, 1024, 80
    B = PR.slaney_mel_basis(sr, n_fft, n).astype(np.float64)
    assert B.shape == (80, 513) and (B >= 0).all()
    df = sr / n_fft
    centres = B.argmax(axis=1) * df
    assert (np.diff(centres) > 0).all()
    area = B.sum(axis=1) * df  # slaney norm: each triangle integrates to ~1 (coarse for the 1-2 bin filters at the bottom)
    assert np.allclose(area[20:], 1.0, atol=0.05), area[20:25]
    low = centres[centres < 900]
    assert np.allclose(np.diff(low), np.diff(low).mean(), atol=df)  # linear spacing below 1 kHz
    assert B[:, 0].sum() == 0 and B[-1].nonzero()[0].max() >= 510  # from fmin = 0 up to Nyquist


def test_kaldi_fbank_vs_the_transformers_kaldi_filter_bank():
    """torchaudio is absent; transformers' SeamlessM4TFeatureExtractor carries an independent implementation of the same Kaldi
    recipe (povey window, pre-emphasis 0.97, DC removal, 80 kaldi-scale mel bins from 20 Hz, log floor 1.19e-7) on samples scaled
    by 2^15 -- `kaldi_fbank(x * 2^15)` must equal its un-normalised features."""
    from transformers import SeamlessM4TFeatureExtractor

    x = torch.randn(1, 16000 * 2 + 123, generator=torch.Generator().manual_seed(3)) * 0.1
    fe = SeamlessM4TFeatureExtractor()
    ref = fe._extract_fbank_features(x[0].numpy())
    got = PR.kaldi_fbank(x * 32768.0).numpy()
    assert got.shape == ref.shape == (1 + (x.shape[1] - 400) // 160, 80)
    assert np.abs(got - ref).max() <= 2e-3 * max(1.0, np.abs(ref).max())
    assert PR.kaldi_fbank(torch.zeros(1, 100)).shape == (0, 80)  # shorter than one window


def test_sinc_resample_properties():
    """torchaudio is absent: the restated Resample is checked on what its algorithm guarantees -- output length
    ceil(n * new / orig), identity at equal rates, a band-limited tone preserved, DC gain 1."""
    sr, n = 22050, 22050
    t = torch.arange(n) / sr
    tone = torch.sin(2 * math.pi * 440.0 * t)[None]
    for new in (16000, 22050, 44100, 8000):
        y = PR.sinc_resample(tone, sr, new)
        assert y.shape == (1, math.ceil(n * new / sr))
        if new == sr:
            assert y is tone
            continue
        want = torch.sin(2 * math.pi * 440.0 * torch.arange(y.shape[1]) / new)[None]
        assert (y - want)[:, 200:-200].abs().max().item() < 2e-3, new
    dc = PR.sinc_resample(torch.ones(2, 3, 4000), 22050, 16000)
    assert dc.shape == (2, 3, math.ceil(4000 * 16000 / 22050)) and (dc[..., 100:-100] - 1).abs().max().item() < 1e-3


def _wav_bytes(x, sr, width=2, channels=1):
    b = io.BytesIO()
    with wave.open(b, "wb") as f:
        f.setnchannels(channels)
        f.setsampwidth(width)
        f.setframerate(sr)
        if width == 2:
            f.writeframes((x * 32767).astype("<i2").tobytes())
        elif width == 1:
            f.writeframes((x * 127 + 128).astype(np.uint8).tobytes())
        elif width == 3:
            v = (x * 8388607).astype(np.int32)
            f.writeframes(b"".join(struct.pack("<i", int(s))[:3] for s in v.reshape(-1)))
        else:
            f.writeframes((x * 2147483647).astype("<i4").tobytes())
    return b.getvalue()


def test_load_and_cut_audio_five_input_forms(tmp_path):
    """infer_v2.py:307-419: path, bytes, (data, sr) with ndarray / Tensor data, bare ndarray / Tensor with sr; mono from the first
    channel for arrays, channel mean for files (librosa.load), truncation to max seconds, the reference's error types."""
    sr = 22050
    x = (0.5 * np.sin(2 * np.pi * 220 * np.arange(sr * 2) / sr)).astype(np.float32)
    raw = _wav_bytes(x, sr)
    p = tmp_path / "a.wav"
    p.write_bytes(raw)
    for src in (str(p), raw):
        a, r = PR.load_and_cut_audio(src, 15)
        assert r == 22050 and a.shape == (1, sr * 2) and a.dtype == torch.float32
        assert (a[0].numpy() - x).__abs__().max() < 1e-4 + 1 / 32767
        a16, r16 = PR.load_and_cut_audio(src, 15, sr=16000)  # the emotion prompt path (infer_v2.py:569)
        assert r16 == 16000 and a16.shape == (1, 32000)
    assert PR.load_and_cut_audio(raw, 1.0)[0].shape == (1, sr)  # cut
    for width in (1, 3, 4):
        a, _ = PR.load_and_cut_audio(_wav_bytes(x, sr, width), 15)
        assert (a[0].numpy() - x).__abs__().max() < (1.5e-2 if width == 1 else 1e-4)
    st = np.stack([x, -x], axis=1).reshape(-1)  # interleaved stereo whose channel mean is silence
    assert PR.load_and_cut_audio(_wav_bytes(st, sr, 2, channels=2), 15)[0].abs().max().item() < 1e-4
    f32 = b"RIFF" + struct.pack("<I", 36 + 4 * x.size) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 3, 1, sr, sr * 4, 4, 32) + b"data" + \
        struct.pack("<I", 4 * x.size) + x.astype("<f4").tobytes()
    assert np.allclose(PR.load_and_cut_audio(f32, 15)[0][0].numpy(), x, atol=1e-6)  # IEEE float WAVE
    a48, r48 = PR.load_and_cut_audio(_wav_bytes(x[:4800], 48000), 15)  # resampled to librosa's default 22050
    assert r48 == 22050 and a48.shape == (1, 2205)
    two = np.stack([x, 0.5 * x])
    for data in (two, torch.from_numpy(two)):
        a, r = PR.load_and_cut_audio((data, 16000), 15)
        assert r == 16000 and a.shape == (1, x.size) and torch.allclose(a[0], torch.from_numpy(x))
        a, r = PR.load_and_cut_audio(data, 15, sr=24000)
        assert r == 24000 and a.shape == (1, x.size)
    assert PR.load_and_cut_audio((x, 16000), 1.5)[0].shape == (1, 24000)
    with pytest.raises(ValueError):
        PR.load_and_cut_audio(x, 15)  # bare array without sr
    with pytest.raises(ValueError):
        PR.load_and_cut_audio(torch.zeros(1, 2, 3), 15, sr=16000)
    with pytest.raises(TypeError):
        PR.load_and_cut_audio(([0.0, 1.0], 16000), 15)
    with pytest.raises(TypeError):
        PR.load_and_cut_audio(12345, 15)
    with pytest.raises(ValueError):
        PR.load_and_cut_audio(b"ID3\x03" + b"\0" * 200, 15)  # not RIFF/WAVE


def test_emotion_matrix_mix_and_normalize():
    """infer_v2.py:552-563,786-792,421-436 against a direct numpy restatement."""
    g = torch.Generator().manual_seed(8)
    emo_num = [3, 5, 2, 4, 1, 2, 3, 6]
    D = 16
    emo = torch.split(torch.randn(sum(emo_num), D, generator=g), emo_num)
    spk = torch.split(torch.randn(sum(emo_num), 12, generator=g), emo_num)
    style = torch.randn(1, 12, generator=g)
    w = [0.0, 0.3, 0.0, 0.1, 0.0, 0.0, 0.25, 0.05]
    mat, ws = PR.emo_vector_mix(w, style, emo, spk, emo_num)
    want = np.zeros(D)
    for k in range(8):
        S = spk[k].numpy()
        cos = S @ style[0].numpy() / (np.linalg.norm(S, axis=1) * np.linalg.norm(style[0].numpy()))
        want += w[k] * emo[k][int(cos.argmax())].numpy()
    assert mat.shape == (1, D) and np.allclose(mat[0].numpy(), want, atol=1e-6) and abs(float(ws) - sum(w)) < 1e-6
    import random

    m2, _ = PR.emo_vector_mix(w, style, emo, spk, emo_num, use_random=True, rng=random.Random(1))
    assert m2.shape == (1, D)
    assert PR.normalize_emo_vec([1.0] + [0.0] * 7) == [0.8] + [0.0] * 7  # biased 0.9375 then capped at 0.8 total
    assert np.allclose(PR.normalize_emo_vec([0.2, 0.2, 0, 0, 0, 0, 0, 0.2]), [0.1875, 0.175, 0, 0, 0, 0, 0, 0.1125])
    assert PR.normalize_emo_vec([0.5, 0.5] + [0.0] * 6, apply_bias=False) == [0.4, 0.4] + [0.0] * 6


def test_prompt_encoder_from_model_dir_files(tmp_path):
    """The file loaders + the once-per-prompt stages end to end on the CPU, from a synthetic model_dir (tests/synthetic_model_dir.py):
    shapes and frame counts as infer_v2.py:508-545 produces them for a 1.5 s prompt recorded at 24 kHz."""
    import os

    from safetensors.torch import load_file

    import synthetic_model_dir as SM
    import voice_tts_amd.s2mel as S2
    from voice_tts_amd.infer_v2 import _s2mel_cfg_from_yaml, _same_prompt, load_s2mel_checkpoint

    root = str(tmp_path / "m")
    _, cfg = SM.write_model_dir(root)
    w2v = PR.W2vBert.from_dir(os.path.join(root, "w2v-bert-2.0"), os.path.join(root, cfg["w2v_stat"]))
    assert w2v.extractor.padding_value == 1 and w2v.extractor.stride == 2
    codec = PR.SemanticCodec(load_file(os.path.join(root, "semantic_codec", "model.safetensors")), cfg["semantic_codec"])
    cam = PR.CamPlus(torch.load(os.path.join(root, "campplus_cn_common.bin"), weights_only=True))
    sd = load_s2mel_checkpoint(os.path.join(root, cfg["s2mel_checkpoint"]))
    assert not any(k.startswith("cfm.module.") for k in sd) and "cfm.estimator.x_embedder.weight_v" not in sd
    sd.update(codec.s2mel_quantizer_tensors())
    scfg = _s2mel_cfg_from_yaml(cfg["s2mel"], S2.S2MEL_CFG)
    scfg.update(codebook_size=8194, codebook_dim=8, semantic_dim=SM.HID)
    s2 = S2.S2Mel(sd, scfg, device="cpu")
    enc = PR.PromptEncoder(w2v, codec, cam, s2, "cpu")
    raw = SM.synthetic_wav_bytes(1.5, 24000)
    spk = enc.speaker(raw)
    n22, n16 = math.ceil(33075 * 16000 / 22050), 33075  # librosa-style load lands on 22050 Hz first (infer_v2.py:331-346)
    frames16 = (1 + (n22 - 400) // 160) // 2  # stride-2 stacking of the 10 ms frames
    Tr = 1 + (n16 + 768 - 1024) // 256  # reflect pad (1024-256)/2 each side, center=False
    assert spk["spk_cond_emb"].shape == (1, frames16, SM.HID) and spk["style"].shape == (1, 192)
    assert spk["ref_mel"].shape == (1, 80, Tr) and spk["prompt_condition"].shape == (1, Tr, scfg["content_dim"])
    assert all(torch.isfinite(v).all() for v in spk.values())
    emo = enc.emotion(raw)
    assert emo.shape[0] == 1 and emo.shape[2] == SM.HID and abs(emo.shape[1] - frames16) <= 1
    # the speaker cache key compares by VALUE, as the reference's `!=` does (a fresh bytes object per request must hit)
    assert _same_prompt(raw, bytes(bytearray(raw))) and _same_prompt("a.wav", "a" + ".wav") and not _same_prompt(raw, raw[:-2])
    x = np.arange(4, dtype=np.float32)
    assert _same_prompt((x, 16000), (x.copy(), 16000)) and not _same_prompt((x, 16000), (x, 22050)) and not _same_prompt(x, torch.from_numpy(x))
