"""Prompt-side glue of `IndexTTS2.infer` (PyTorch-ROCm hosted, as `north_star` leaves it): everything the reference does
ONCE per new speaker / emotion prompt before the segment loop (indextts/infer_v2.py:508-580), so that
`IndexTTS2(cfg_path, model_dir).infer(wav, text)` runs without an injected `Glue`.

  infer_v2.py:307-419      _load_and_cut_audio: the five `spk_audio_prompt` forms (path, bytes, (data, sr), ndarray, Tensor)
  infer_v2.py:515-517      torchaudio.transforms.Resample(sr, 22050 / 16000)           -> `sinc_resample`
  infer_v2.py:519-524,200-209  SeamlessM4TFeatureExtractor + Wav2Vec2BertModel hidden_states[17], (x - mean) / std -> `W2vBert`
  infer_v2.py:526          semantic_codec.quantize (repcodec_model.py:176-196, vocos.py:468-526,719-782,
                           factorized_vector_quantize.py:49-127, residual_vq.py:75-141)  -> `SemanticCodec`
  infer_v2.py:527          mel_fn = s2mel/modules/audio.py:45-82 (reflect pad, hann STFT, slaney mel basis, log clamp 1e-5) -> `mel_spectrogram`
  infer_v2.py:529-534      torchaudio.compliance.kaldi.fbank(num_mel_bins=80, dither=0) - mean, CAMPPlus
                           (campplus/DTDNN.py:14-115, layers.py)                        -> `kaldi_fbank`, `CamPlus`
  infer_v2.py:536-539      prompt_condition = length_regulator(S_ref, ylens = ref_mel frames)  (s2mel.py's regulator)
  infer_v2.py:552-563,786-792  emotion-matrix mix (`find_most_similar_cosine` over spk_matrix, weighted rows of emo_matrix)
  infer_v2.py:421-436      normalize_emo_vec

What is pinned and what is not: `SemanticCodec`, `CamPlus`, the STFT / log part of `mel_spectrogram` and
the emotion mix are held to fixtures produced by the reference's own classes (tests/golden/prompt_tiny.npz).  librosa (file
decoding + soxr resampling, slaney mel basis) and torchaudio (sinc resampler, kaldi fbank) are absent from this image: those
four are restatements of the libraries' published algorithms, "parity unpinned" by library output; `kaldi_fbank` is
cross-checked against the independent Kaldi-style filter bank inside transformers' SeamlessM4TFeatureExtractor.  w2v-bert-2.0
itself runs through the installed `transformers` classes from a LOCAL directory only (nothing is ever fetched).
"""
import io
import math
import os
import struct

import numpy as np
import torch
import torch.nn.functional as F

from .convs import conv1d, conv2d  # GEMM forms: no MIOpen in the request path (convs.py)

from .weights import fold_weight_norm


# ===================================================================================================== audio decode
class UnsupportedAudioError(ValueError):
    """Speaker / emotion audio the built-in decoder cannot read.  The wire contract of this build: prompt
    audio as a path or a byte string is RIFF/WAVE -- PCM 8/16/24/32-bit or IEEE float 32/64, any channel count and rate.
    (The reference decodes through `librosa.load`, i.e. soundfile / audioread: mp3, flac, ogg too; those libraries are absent
    here and nothing is fetched.)  `/tts` answers 415 for it, naming the supported container."""


def _decode_riff(buf):
    """RIFF/WAVE bytes -> (float32 [channels, samples] in [-1, 1), sr).  PCM 8/16/24/32-bit, IEEE float 32/64, and
    WAVE_FORMAT_EXTENSIBLE wrappers of those (what `librosa.load` -> soundfile returns as float32)."""
    if len(buf) < 12 or buf[:4] != b"RIFF" or buf[8:12] != b"WAVE":
        raise UnsupportedAudioError("prompt audio is not a RIFF/WAVE stream: this build decodes WAV only (mp3 / flac / ogg need a decoder "
                                    "library the image does not have); send 8/16/24/32-bit PCM or 32/64-bit float WAV")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(buf):
        cid, size = buf[pos:pos + 4], struct.unpack("<I", buf[pos + 4:pos + 8])[0]
        body = buf[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            if len(body) < 16:
                raise UnsupportedAudioError(f"WAVE stream with a truncated fmt chunk ({len(body)} bytes, 16 needed)")
            tag, ch, sr, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE and len(body) >= 26:  # extensible: the real tag is the first 2 bytes of the sub-format GUID
                tag = struct.unpack("<H", body[24:26])[0]
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            data = body
        pos += 8 + size + (size & 1)
    if fmt is None or data is None:
        raise UnsupportedAudioError("WAVE stream without fmt/data chunk")
    tag, ch, sr, bits = fmt
    if ch < 1 or sr < 1:
        raise UnsupportedAudioError(f"WAVE header with {ch} channels at {sr} Hz")
    if (tag == 1 and bits not in (8, 16, 24, 32)) or (tag == 3 and bits not in (32, 64)):
        raise UnsupportedAudioError(f"WAVE format tag {tag} with {bits}-bit samples is not supported (PCM 8/16/24/32, float 32/64)")
    if tag == 1:
        if bits == 8:
            x = (np.frombuffer(data, np.uint8).astype(np.float32) - 128.0) / 128.0
        elif bits == 16:
            x = np.frombuffer(data[: len(data) // 2 * 2], "<i2").astype(np.float32) / 32768.0
        elif bits == 24:
            b = np.frombuffer(data[: len(data) // 3 * 3], np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            x = ((v ^ 0x800000) - 0x800000).astype(np.float32) / 8388608.0
        elif bits == 32:
            x = (np.frombuffer(data[: len(data) // 4 * 4], "<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    elif tag == 3:
        x = np.frombuffer(data[: len(data) // (bits // 8) * (bits // 8)], "<f4" if bits == 32 else "<f8").astype(np.float32)
    else:
        raise UnsupportedAudioError(f"WAVE format tag {tag} is not supported (1 = PCM, 3 = IEEE float, or their EXTENSIBLE wrappers)")
    n = x.size // ch
    return x[: n * ch].reshape(n, ch).T.copy(), sr


def _poly_resample(x, sr_in, sr_out):
    """Stand-in for librosa's default `res_type="soxr_hq"` (soxr is absent): scipy's polyphase Kaiser FIR resampler."""
    if sr_in == sr_out:
        return x
    from scipy.signal import resample_poly

    g = math.gcd(int(sr_in), int(sr_out))
    return resample_poly(x.astype(np.float64), int(sr_out) // g, int(sr_in) // g).astype(np.float32)


def librosa_style_load(src, sr=22050):
    """`librosa.load(src, sr=sr)` for a path or a file-like of WAVE data: float32, mono (channel mean), resampled to `sr`
    (librosa's default is 22050, which is what `_load_and_cut_audio` gets when it passes no sr; infer_v2.py:331-346)."""
    if isinstance(src, (bytes, bytearray)):
        buf = bytes(src)
    elif hasattr(src, "read"):
        buf = src.read()
    else:
        with open(src, "rb") as f:
            buf = f.read()
    x, sr_in = _decode_riff(buf)
    mono = x.mean(axis=0) if x.shape[0] > 1 else x[0]
    return _poly_resample(mono, sr_in, sr), sr


def load_and_cut_audio(audio_input, max_audio_length_seconds, sr=None):
    """`IndexTTS2._load_and_cut_audio` (infer_v2.py:307-419): -> (float tensor [1, samples], sample rate)."""
    if isinstance(audio_input, (str, os.PathLike)) or isinstance(audio_input, (bytes, bytearray)):
        src = io.BytesIO(bytes(audio_input)) if isinstance(audio_input, (bytes, bytearray)) else audio_input
        audio, sr = librosa_style_load(src, sr=sr if sr else 22050)
        audio = torch.tensor(audio).unsqueeze(0)
    elif isinstance(audio_input, tuple):
        audio_data, input_sr = audio_input
        if isinstance(audio_data, np.ndarray):
            audio = torch.from_numpy(audio_data).float()
        elif isinstance(audio_data, torch.Tensor):
            audio = audio_data.float()
        else:
            raise TypeError(f"Unsupported audio_data type in tuple: {type(audio_data)}")
        if audio.dim() == 1:
            audio = audio.unsqueeze(0)
        elif audio.dim() > 2:
            raise ValueError(f"Audio tensor has too many dimensions: {audio.dim()}")
        if audio.shape[0] > 1:
            audio = audio[0:1, :]
        sr = input_sr
    elif isinstance(audio_input, (np.ndarray, torch.Tensor)):
        kind = "numpy.ndarray" if isinstance(audio_input, np.ndarray) else "torch.Tensor"
        if sr is None:
            raise ValueError(f"Sample rate (sr) must be provided when passing {kind}")
        audio = (torch.from_numpy(audio_input) if isinstance(audio_input, np.ndarray) else audio_input).float()
        if audio.dim() == 1:
            audio = audio.unsqueeze(0)
        elif audio.dim() > 2:
            raise ValueError(f"Audio {'array' if kind.startswith('numpy') else 'tensor'} has too many dimensions: {audio.dim()}")
        if audio.shape[0] > 1:
            audio = audio[0:1, :]
    else:
        raise TypeError(f"Unsupported audio_input type: {type(audio_input)}. Expected str, bytes, tuple, numpy.ndarray, or torch.Tensor")
    max_audio_samples = int(max_audio_length_seconds * sr)
    if audio.shape[1] > max_audio_samples:
        audio = audio[:, :max_audio_samples]
    return audio, sr


# ===================================================================================================== resampling
_RESAMPLE_KERNELS = {}


def sinc_resample(wave, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """`torchaudio.transforms.Resample(orig, new)` with its defaults (sinc_interp_hann): one Hann-windowed sinc kernel per
    output phase, applied as a strided conv1d.  wave [..., time] -> [..., ceil(time * new / orig)]."""
    orig_freq, new_freq = int(orig_freq), int(new_freq)
    if orig_freq == new_freq:
        return wave
    g = math.gcd(orig_freq, new_freq)
    orig, new = orig_freq // g, new_freq // g
    key = (orig, new, lowpass_filter_width, rolloff, wave.device)
    if key not in _RESAMPLE_KERNELS:
        base = min(orig, new) * rolloff
        width = math.ceil(lowpass_filter_width * orig / base)
        idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
        t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
        t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
        window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
        t = t * math.pi
        kern = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / orig)
        _RESAMPLE_KERNELS[key] = (kern.to(torch.float32).to(wave.device), width)
    kern, width = _RESAMPLE_KERNELS[key]
    shape = wave.shape
    x = wave.reshape(-1, shape[-1]).float()
    x = F.pad(x, (width, width + orig))
    y = conv1d(x[:, None], kern, stride=orig).transpose(1, 2).reshape(x.shape[0], -1)
    target = int(math.ceil(new * shape[-1] / orig))
    return y[..., :target].reshape(shape[:-1] + (target,))


# ===================================================================================================== reference mel (22.05 kHz)
def slaney_mel_basis(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    """`librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)` with its defaults (htk=False, norm="slaney"): triangles on the
    Slaney auditory scale (linear below 1 kHz, log above), each normalised to unit area in Hz.  -> float32 [n_mels, 1 + n_fft//2]."""
    fmax = sr / 2.0 if fmax is None else float(fmax)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, math.log(6.4) / 27.0

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    weights = np.maximum(0, np.minimum(lower, upper))
    weights *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return weights.astype(np.float32)


_MEL_CACHE = {}


def mel_spectrogram(y, n_fft=1024, num_mels=80, sampling_rate=22050, hop_size=256, win_size=1024, fmin=0, fmax=None, center=False,
                    mel_basis=None):
    """s2mel/modules/audio.py:45-82.  y [B, samples] -> log-mel [B, num_mels, frames]."""
    key = (sampling_rate, n_fft, num_mels, fmin, fmax, win_size, str(y.device))
    if key not in _MEL_CACHE:
        _MEL_CACHE[key] = (torch.from_numpy(slaney_mel_basis(sampling_rate, n_fft, num_mels, fmin, fmax)).to(y.device),
                           torch.hann_window(win_size).to(y.device))
    basis, window = _MEL_CACHE[key]
    if mel_basis is not None:
        basis = mel_basis.to(y.device)
    p = int((n_fft - hop_size) / 2)
    y = F.pad(y.unsqueeze(1), (p, p), mode="reflect").squeeze(1)
    spec = torch.view_as_real(torch.stft(y, n_fft, hop_length=hop_size, win_length=win_size, window=window, center=center, pad_mode="reflect",
                                         normalized=False, onesided=True, return_complex=True))
    spec = torch.sqrt(spec.pow(2).sum(-1) + 1e-9)
    return torch.log(torch.clamp(torch.matmul(basis, spec), min=1e-5))


# ===================================================================================================== kaldi fbank (16 kHz)
_FBANK_CACHE = {}


def kaldi_fbank(waveform, num_mel_bins=80, sample_frequency=16000.0, frame_length=25.0, frame_shift=10.0, preemphasis=0.97,
                low_freq=20.0, high_freq=0.0):
    """`torchaudio.compliance.kaldi.fbank(waveform, num_mel_bins=80, dither=0, sample_frequency=16000)` with the remaining
    defaults (snip_edges, remove_dc_offset, povey window, round_to_power_of_two, use_power, use_log_fbank, no energy):
    Kaldi's `compute-fbank-feats`.  waveform [1, samples] -> [frames, num_mel_bins]."""
    x = waveform[0].float()
    win, shift = int(sample_frequency * frame_length * 0.001), int(sample_frequency * frame_shift * 0.001)
    nfft = 1 << (win - 1).bit_length()
    if x.numel() < win:
        return x.new_zeros(0, num_mel_bins)
    m = 1 + (x.numel() - win) // shift
    frames = x.unfold(0, win, shift)[:m]
    frames = frames - frames.mean(dim=1, keepdim=True)  # remove_dc_offset
    prev = torch.cat((frames[:, :1], frames[:, :-1]), dim=1)  # replicate-padded shift
    frames = frames - preemphasis * prev
    key = (win, nfft, num_mel_bins, sample_frequency, low_freq, high_freq, str(x.device))
    if key not in _FBANK_CACHE:
        window = torch.hann_window(win, periodic=False, dtype=torch.float64).pow(0.85)  # povey
        nyq = 0.5 * sample_frequency
        hi = high_freq + nyq if high_freq <= 0 else high_freq
        mel = lambda f: 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)
        lo_m, hi_m = mel(low_freq), mel(hi)
        delta = (hi_m - lo_m) / (num_mel_bins + 1)
        b = np.arange(num_mel_bins, dtype=np.float64)[:, None]
        left, center, right = lo_m + b * delta, lo_m + (b + 1) * delta, lo_m + (b + 2) * delta
        fm = mel(sample_frequency / nfft * np.arange(nfft // 2, dtype=np.float64))[None, :]
        bank = np.maximum(0.0, np.minimum((fm - left) / (center - left), (right - fm) / (right - center)))
        bank = np.pad(bank, ((0, 0), (0, 1)))  # the Nyquist bin carries no weight
        _FBANK_CACHE[key] = (window.float().to(x.device), torch.from_numpy(bank).float().to(x.device))
    window, bank = _FBANK_CACHE[key]
    frames = F.pad(frames * window, (0, nfft - win))
    power = torch.fft.rfft(frames).abs().pow(2.0)
    return torch.clamp(power @ bank.t(), min=1.1920928955078125e-07).log()


# ===================================================================================================== CAM++ (style vector)
class CamPlus:
    """CAMPPlus(feat_dim=80, embedding_size=192) in eval mode (campplus/DTDNN.py:50-115; FCM head :14-48; layers.py).
    `W` keyed like `CAMPPlus.state_dict()` (campplus_cn_common.bin)."""

    BLOCKS = ((12, 3, 1), (24, 3, 2), (16, 3, 2))  # (layers, kernel, dilation)

    def __init__(self, W, device="cpu"):
        self.W = {k: v.to(device, torch.float32) for k, v in W.items() if v.is_floating_point()}

    def _bn(self, x, p, affine=True):
        W = self.W
        return F.batch_norm(x, W[p + "running_mean"], W[p + "running_var"], W.get(p + "weight") if affine else None,
                            W.get(p + "bias") if affine else None, False, 0.0, 1e-5)

    def _res_block(self, x, p, stride):
        W = self.W
        out = F.relu(self._bn(conv2d(x, W[p + "conv1.weight"], None, (stride, 1), 1), p + "bn1."))
        out = self._bn(conv2d(out, W[p + "conv2.weight"], None, 1, 1), p + "bn2.")
        if p + "shortcut.0.weight" in W:
            x = self._bn(conv2d(x, W[p + "shortcut.0.weight"], None, (stride, 1)), p + "shortcut.1.")
        return F.relu(out + x)

    @staticmethod
    def _seg_pool(x, seg_len=100):
        seg = F.avg_pool1d(x, kernel_size=seg_len, stride=seg_len, ceil_mode=True)
        return seg.unsqueeze(-1).expand(*seg.shape, seg_len).reshape(*seg.shape[:-1], -1)[..., : x.shape[-1]]

    @torch.no_grad()
    def __call__(self, feat):
        """feat [B, T, 80] (mean-normalised fbank) -> style [B, embedding_size]."""
        W = self.W
        x = feat.permute(0, 2, 1).unsqueeze(1)  # [B, 1, F, T]
        x = F.relu(self._bn(conv2d(x, W["head.conv1.weight"], None, 1, 1), "head.bn1."))
        for layer in ("layer1", "layer2"):
            for i in range(2):
                x = self._res_block(x, f"head.{layer}.{i}.", 2 if i == 0 else 1)
        x = F.relu(self._bn(conv2d(x, W["head.conv2.weight"], None, (2, 1), 1), "head.bn2."))
        x = x.reshape(x.shape[0], x.shape[1] * x.shape[2], x.shape[3])
        x = F.relu(self._bn(conv1d(x, W["xvector.tdnn.linear.weight"], None, 2, 2), "xvector.tdnn.nonlinear.batchnorm."))
        for bi, (n_layers, k, dil) in enumerate(self.BLOCKS, start=1):
            for li in range(1, n_layers + 1):
                p = f"xvector.block{bi}.tdnnd{li}."
                h = conv1d(F.relu(self._bn(x, p + "nonlinear1.batchnorm.")), W[p + "linear1.weight"])
                h = F.relu(self._bn(h, p + "nonlinear2.batchnorm."))
                y = conv1d(h, W[p + "cam_layer.linear_local.weight"], None, 1, (k - 1) // 2 * dil, dil)
                ctx = h.mean(-1, keepdim=True) + self._seg_pool(h)
                ctx = F.relu(conv1d(ctx, W[p + "cam_layer.linear1.weight"], W[p + "cam_layer.linear1.bias"]))
                gate = torch.sigmoid(conv1d(ctx, W[p + "cam_layer.linear2.weight"], W[p + "cam_layer.linear2.bias"]))
                x = torch.cat([x, y * gate], dim=1)
            p = f"xvector.transit{bi}."
            x = conv1d(F.relu(self._bn(x, p + "nonlinear.batchnorm.")), W[p + "linear.weight"])
        x = F.relu(self._bn(x, "xvector.out_nonlinear.batchnorm."))
        stats = torch.cat([x.mean(dim=-1), x.std(dim=-1, unbiased=True)], dim=-1)
        out = conv1d(stats.unsqueeze(-1), W["xvector.dense.linear.weight"]).squeeze(-1)
        return self._bn(out, "xvector.dense.nonlinear.batchnorm.", affine=False)


def camplus_shapes(feat_dim=80, embedding_size=192, growth=32, bn_size=4, init_channels=128, m_channels=32):
    """(name, shape) of every CAMPPlus tensor (for seeded twins in tests / bench)."""
    out = []

    def bn(p, c, affine=True):
        if affine:
            out.extend([(p + "weight", (c,)), (p + "bias", (c,))])
        out.extend([(p + "running_mean", (c,)), (p + "running_var", (c,))])

    out.append(("head.conv1.weight", (m_channels, 1, 3, 3)))
    bn("head.bn1.", m_channels)
    for layer in ("layer1", "layer2"):
        for i in range(2):
            p = f"head.{layer}.{i}."
            out.append((p + "conv1.weight", (m_channels, m_channels, 3, 3)))
            bn(p + "bn1.", m_channels)
            out.append((p + "conv2.weight", (m_channels, m_channels, 3, 3)))
            bn(p + "bn2.", m_channels)
            if i == 0:
                out.append((p + "shortcut.0.weight", (m_channels, m_channels, 1, 1)))
                bn(p + "shortcut.1.", m_channels)
    out.append(("head.conv2.weight", (m_channels, m_channels, 3, 3)))
    bn("head.bn2.", m_channels)
    ch = m_channels * (feat_dim // 8)
    out.append(("xvector.tdnn.linear.weight", (init_channels, ch, 5)))
    bn("xvector.tdnn.nonlinear.batchnorm.", init_channels)
    ch = init_channels
    for bi, (n_layers, k, dil) in enumerate(CamPlus.BLOCKS, start=1):
        for li in range(1, n_layers + 1):
            p = f"xvector.block{bi}.tdnnd{li}."
            cin, bnc = ch + (li - 1) * growth, bn_size * growth
            bn(p + "nonlinear1.batchnorm.", cin)
            out.append((p + "linear1.weight", (bnc, cin, 1)))
            bn(p + "nonlinear2.batchnorm.", bnc)
            out.extend([(p + "cam_layer.linear_local.weight", (growth, bnc, k)), (p + "cam_layer.linear1.weight", (bnc // 2, bnc, 1)),
                        (p + "cam_layer.linear1.bias", (bnc // 2,)), (p + "cam_layer.linear2.weight", (growth, bnc // 2, 1)),
                        (p + "cam_layer.linear2.bias", (growth,))])
        ch = ch + n_layers * growth
        bn(f"xvector.transit{bi}.nonlinear.batchnorm.", ch)
        out.append((f"xvector.transit{bi}.linear.weight", (ch // 2, ch, 1)))
        ch //= 2
    bn("xvector.out_nonlinear.batchnorm.", ch)
    out.append(("xvector.dense.linear.weight", (embedding_size, ch * 2, 1)))
    bn("xvector.dense.nonlinear.batchnorm.", embedding_size, affine=False)
    return out


def make_camplus_weights(seed=1234, **kw):
    g = torch.Generator().manual_seed(seed)
    W = {}
    for name, shape in camplus_shapes(**kw):
        if name.endswith("running_var"):
            W[name] = 0.5 + torch.rand(shape, generator=g)
        elif name.endswith("running_mean") or name.endswith("bias"):
            W[name] = 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("batchnorm.weight") or name.endswith("bn1.weight") or name.endswith("bn2.weight") or name.endswith("shortcut.1.weight"):
            W[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            fan_in = int(np.prod(shape[1:]))
            W[name] = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
    return W


# ===================================================================================================== semantic codec (RepCodec.quantize)
CODEC_CFG = dict(codebook_size=8192, hidden_size=1024, codebook_dim=8, vocos_dim=384, vocos_intermediate_dim=2048, vocos_num_layers=12)


class SemanticCodec:
    """`RepCodec.quantize` (repcodec_model.py:176-196) with one factorized quantizer and no down-sampling (the MaskGCT semantic
    codec): VocosBackbone encoder (k=7 embed conv, LayerNorm, ConvNeXt blocks, final LayerNorm) -> Linear -> in_project ->
    nearest codebook row by cosine -> out_project.  `W` keyed like `RepCodec.state_dict()` (semantic_codec/model.safetensors);
    only `encoder.*` and `quantizer.quantizers.0.*` are read."""

    def __init__(self, W, cfg=CODEC_CFG, device="cpu"):
        W = fold_weight_norm({k: v for k, v in W.items() if k.startswith(("encoder.", "quantizer."))})
        self.W = {k: v.to(device, torch.float32) for k, v in W.items()}
        self.cfg = dict(cfg)

    @torch.no_grad()
    def quantize(self, x):
        """x [B, T, hidden] (normalised w2v-bert features) -> (codes [B, T] int64, S_ref [B, T, hidden])."""
        W, q = self.W, "quantizer.quantizers.0."
        h = conv1d(x.transpose(1, 2), W["encoder.0.embed.weight"], W["encoder.0.embed.bias"], padding=3)
        h = F.layer_norm(h.transpose(1, 2), h.shape[1:2], W["encoder.0.norm.weight"], W["encoder.0.norm.bias"], 1e-6).transpose(1, 2)
        for i in range(self.cfg["vocos_num_layers"]):
            p = f"encoder.0.convnext.{i}."
            r = conv1d(h, W[p + "dwconv.weight"], W[p + "dwconv.bias"], padding=3, groups=h.shape[1]).transpose(1, 2)
            r = F.layer_norm(r, r.shape[-1:], W[p + "norm.weight"], W[p + "norm.bias"], 1e-6)
            r = F.linear(F.gelu(F.linear(r, W[p + "pwconv1.weight"], W[p + "pwconv1.bias"])), W[p + "pwconv2.weight"], W[p + "pwconv2.bias"])
            if p + "gamma" in W:
                r = W[p + "gamma"] * r
            h = h + r.transpose(1, 2)
        h = F.layer_norm(h.transpose(1, 2), h.shape[1:2], W["encoder.0.final_layer_norm.weight"], W["encoder.0.final_layer_norm.bias"], 1e-6)
        z = F.linear(h, W["encoder.1.weight"], W["encoder.1.bias"])  # [B, T, hidden]
        z_e = F.linear(z, W[q + "in_project.weight"].squeeze(-1), W[q + "in_project.bias"])  # 1x1 conv
        enc = F.normalize(z_e.reshape(-1, z_e.shape[-1]))
        cb = F.normalize(W[q + "codebook.weight"])
        dist = enc.pow(2).sum(1, keepdim=True) - 2 * enc @ cb.t() + cb.pow(2).sum(1, keepdim=True).t()
        idx = (-dist).max(1)[1].reshape(z_e.shape[0], z_e.shape[1])
        z_q = F.embedding(idx, W[q + "codebook.weight"])
        out = F.linear(z_q, W[q + "out_project.weight"].squeeze(-1), W[q + "out_project.bias"])
        return idx, out

    def s2mel_quantizer_tensors(self):
        """The three tensors `S2Mel.vq2emb` needs (`semantic_codec.quantizer.vq2emb`, infer_v2.py:714), under its key names."""
        q = "quantizer.quantizers.0."
        return {"quantizer.codebook.weight": self.W[q + "codebook.weight"], "quantizer.out_project.weight": self.W[q + "out_project.weight"],
                "quantizer.out_project.bias": self.W[q + "out_project.bias"]}


def codec_shapes(cfg=CODEC_CFG):
    H, D, I, n, cd = cfg["hidden_size"], cfg["vocos_dim"], cfg["vocos_intermediate_dim"], cfg["vocos_num_layers"], cfg["codebook_dim"]
    out = [("encoder.0.embed.weight", (D, H, 7)), ("encoder.0.embed.bias", (D,)), ("encoder.0.norm.weight", (D,)), ("encoder.0.norm.bias", (D,))]
    for i in range(n):
        p = f"encoder.0.convnext.{i}."
        out += [(p + "dwconv.weight", (D, 1, 7)), (p + "dwconv.bias", (D,)), (p + "norm.weight", (D,)), (p + "norm.bias", (D,)),
                (p + "pwconv1.weight", (I, D)), (p + "pwconv1.bias", (I,)), (p + "pwconv2.weight", (D, I)), (p + "pwconv2.bias", (D,)), (p + "gamma", (D,))]
    out += [("encoder.0.final_layer_norm.weight", (D,)), ("encoder.0.final_layer_norm.bias", (D,)), ("encoder.1.weight", (H, D)), ("encoder.1.bias", (H,))]
    q = "quantizer.quantizers.0."
    out += [(q + "in_project.weight", (cd, H, 1)), (q + "in_project.bias", (cd,)), (q + "out_project.weight", (H, cd, 1)), (q + "out_project.bias", (H,)),
            (q + "codebook.weight", (cfg["codebook_size"], cd))]
    return out


def make_codec_weights(cfg=CODEC_CFG, seed=1234):
    g = torch.Generator().manual_seed(seed)
    W = {}
    for name, shape in codec_shapes(cfg):
        if name.endswith("norm.weight"):
            W[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("gamma"):
            W[name] = torch.full(shape, 1.0 / cfg["vocos_num_layers"]) * (1.0 + 0.1 * torch.randn(shape, generator=g))
        elif name.endswith("bias"):
            W[name] = 0.02 * torch.randn(shape, generator=g)
        elif name.endswith("codebook.weight"):
            W[name] = torch.randn(shape, generator=g)
        else:
            W[name] = torch.randn(shape, generator=g) / math.sqrt(int(np.prod(shape[1:])))
    return W


# ===================================================================================================== w2v-bert features
class W2vBert:
    """SeamlessM4TFeatureExtractor + Wav2Vec2BertModel hidden_states[17], normalised with the MaskGCT statistics
    (infer_v2.py:114-120,200-209; maskgct_utils.py:87-93).  Both are the installed `transformers` classes -- third-party
    pretrained components the reference also takes from that library -- built from LOCAL files only."""

    def __init__(self, model, mean, std, extractor=None, device="cpu", layer=17):
        from transformers import SeamlessM4TFeatureExtractor

        self.device = torch.device(device)
        # hidden_states[layer] is the INPUT of encoder layer `layer` (the encoder appends the running state before each layer and once
        # after the loop; no final norm): layers `layer` .. 23 never reach what is read.  The reference runs all 24 and reads [17];
        # dropping the unread ones returns the same tensor for 17/24 of the passes (and 0.7 GB less on the device).
        enc = getattr(model, "encoder", None)
        if enc is not None and hasattr(enc, "layers") and len(enc.layers) > layer:
            enc.layers = enc.layers[:layer]
        self.model = model.to(self.device).eval()
        # preprocessor_config.json of facebook/w2v-bert-2.0: 80 mel bins, stride 2, 16 kHz, padding value 1
        self.extractor = extractor if extractor is not None else SeamlessM4TFeatureExtractor(feature_size=80, num_mel_bins=80, sampling_rate=16000,
                                                                                             stride=2, padding_value=1)
        self.mean, self.std = mean.to(self.device, torch.float32), std.to(self.device, torch.float32)
        self.layer = layer

    @classmethod
    def from_dir(cls, path, stats_path, device="cpu"):
        from transformers import SeamlessM4TFeatureExtractor, Wav2Vec2BertModel

        if not os.path.isdir(path):
            raise FileNotFoundError(f"w2v-bert-2.0 directory {path} not found (config.json + model.safetensors + preprocessor_config.json; nothing is downloaded)")
        model = Wav2Vec2BertModel.from_pretrained(path, local_files_only=True)
        ex = SeamlessM4TFeatureExtractor.from_pretrained(path, local_files_only=True) if os.path.isfile(os.path.join(path, "preprocessor_config.json")) else None
        st = torch.load(stats_path, map_location="cpu", weights_only=True)
        return cls(model, st["mean"], torch.sqrt(st["var"]), ex, device)

    @torch.no_grad()
    def __call__(self, audio_16k):
        """audio [1, samples] at 16 kHz (host tensor) -> [1, T, hidden] on the device."""
        inputs = self.extractor(audio_16k, sampling_rate=16000, return_tensors="pt")
        out = self.model(input_features=inputs["input_features"].to(self.device), attention_mask=inputs["attention_mask"].to(self.device),
                         output_hidden_states=True)
        return (out.hidden_states[self.layer] - self.mean) / self.std


# ===================================================================================================== emotion matrix
def find_most_similar_cosine(query_vector, matrix):
    """infer_v2.py:786-792."""
    return torch.argmax(F.cosine_similarity(query_vector.float(), matrix.float(), dim=1))


def normalize_emo_vec(emo_vector, apply_bias=True):
    """`IndexTTS2.normalize_emo_vec` (infer_v2.py:421-436)."""
    if apply_bias:
        emo_bias = [0.9375, 0.875, 1.0, 1.0, 0.9375, 0.9375, 0.6875, 0.5625]
        emo_vector = [vec * bias for vec, bias in zip(emo_vector, emo_bias)]
    emo_sum = sum(emo_vector)
    if emo_sum > 0.8:
        scale_factor = 0.8 / emo_sum
        emo_vector = [vec * scale_factor for vec in emo_vector]
    return emo_vector


def emo_vector_mix(emo_vector, style, emo_matrix, spk_matrix, emo_num, use_random=False, rng=None):
    """infer_v2.py:552-563: per emotion group, the row of `emo_matrix` whose speaker row in `spk_matrix` is closest (cosine)
    to this prompt's style vector (or a random row), weighted by the request's 8 strengths.  -> (emovec_mat [1, D], sum of weights)."""
    import random

    weight_vector = torch.tensor(emo_vector, device=style.device)
    if use_random:
        r = rng if rng is not None else random
        index = [r.randint(0, x - 1) for x in emo_num]
    else:
        index = [find_most_similar_cosine(style, tmp) for tmp in spk_matrix]
    rows = torch.cat([tmp[i].unsqueeze(0) for i, tmp in zip(index, emo_matrix)], 0)
    return torch.sum(weight_vector.unsqueeze(1) * rows, 0).unsqueeze(0), torch.sum(weight_vector)


# ===================================================================================================== the default glue
class PromptEncoder:
    """The once-per-prompt stages (infer_v2.py:508-545,565-580) over the models above + the s2mel length regulator."""

    def __init__(self, w2v, codec, camplus, s2mel, device, mel_args=None):
        self.w2v, self.codec, self.camplus, self.s2mel = w2v, codec, camplus, s2mel
        self.device = torch.device(device)
        self.mel_args = dict(n_fft=1024, win_size=1024, hop_size=256, num_mels=80, sampling_rate=22050, fmin=0, fmax=None, center=False)
        if mel_args:
            self.mel_args.update(mel_args)

    @torch.no_grad()
    def speaker(self, spk_audio_prompt):
        audio, sr = load_and_cut_audio(spk_audio_prompt, 15)
        audio_22k = sinc_resample(audio, sr, 22050)
        audio_16k = sinc_resample(audio, sr, 16000)
        spk_cond_emb = self.w2v(audio_16k)
        _, S_ref = self.codec.quantize(spk_cond_emb)
        ref_mel = mel_spectrogram(audio_22k.to(self.device).float(), **self.mel_args)
        feat = kaldi_fbank(audio_16k.to(self.device), num_mel_bins=80, sample_frequency=16000)
        feat = feat - feat.mean(dim=0, keepdim=True)
        style = self.camplus(feat.unsqueeze(0))
        prompt_condition = self.s2mel.length_regulator(S_ref, torch.tensor([ref_mel.size(2)], device=self.device))
        return dict(spk_cond_emb=spk_cond_emb, style=style, prompt_condition=prompt_condition, ref_mel=ref_mel)

    @torch.no_grad()
    def emotion(self, emo_audio_prompt):
        emo_audio, _ = load_and_cut_audio(emo_audio_prompt, 15, sr=16000)
        return self.w2v(emo_audio)
