"""Where a NEW speaker prompt's ~80 ms go (bench.py extra.prompt_side / the `prompt` stage of mixed64): per component, over 8
distinct 5 s recordings at production model sizes (random weights)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
import voice_tts_amd.prompt as PR  # noqa: E402
import voice_tts_amd.s2mel as S2  # noqa: E402

dev = torch.device("cuda:0")


class _HP:
    pass


hp = _HP()
hp.s2mel_model = S2.S2Mel(S2.make_s2mel_weights(S2.S2MEL_CFG, seed=1234), S2.S2MEL_CFG, device=dev)
enc = B.build_prompt_encoder(hp, dev)
wavs = [B.prompt_wav(5.0, 24000, seed=i) for i in range(10)]
acc = {}


def tick(name, t0):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    acc.setdefault(name, []).append((t1 - t0) * 1e3)
    return t1


for i, w in enumerate(wavs):
    torch.cuda.synchronize()
    t = time.perf_counter()
    audio, sr = PR.load_and_cut_audio(w, 15)
    t = tick("wav decode + librosa-style resample to 22.05 kHz (CPU)", t)
    a22 = PR.sinc_resample(audio, sr, 22050)
    a16 = PR.sinc_resample(audio, sr, 16000)
    t = tick("sinc resample to 22.05 / 16 kHz (CPU)", t)
    inputs = enc.w2v.extractor(a16, sampling_rate=16000, return_tensors="pt")
    t = tick("SeamlessM4T feature extractor (CPU, numpy)", t)
    with torch.no_grad():
        out = enc.w2v.model(input_features=inputs["input_features"].to(dev), attention_mask=inputs["attention_mask"].to(dev), output_hidden_states=True)
        emb = (out.hidden_states[enc.w2v.layer] - enc.w2v.mean) / enc.w2v.std
    t = tick("w2v-bert-2.0-shaped encoder forward (GPU, transformers)", t)
    with torch.no_grad():
        _, S_ref = enc.codec.quantize(emb)
        t = tick("semantic codec quantize", t)
        ref_mel = PR.mel_spectrogram(a22.to(dev).float(), **enc.mel_args)
        t = tick("reference mel", t)
        feat = PR.kaldi_fbank(a16.to(dev), num_mel_bins=80, sample_frequency=16000)
        feat = feat - feat.mean(dim=0, keepdim=True)
        style = enc.camplus(feat.unsqueeze(0))
        t = tick("kaldi fbank + CAM++", t)
        pc = enc.s2mel.length_regulator(S_ref, torch.tensor([ref_mel.size(2)], device=dev))
        t = tick("prompt condition (length regulator)", t)
tot = 0.0
for k, v in acc.items():
    v = sorted(v[2:])  # the first two requests pay the one-off costs
    med = v[len(v) // 2]
    tot += med
    print(f"{k:62s} median {med:7.2f} ms  (min {v[0]:.2f}, max {v[-1]:.2f}; first request {acc[k][0]:.1f})")
print(f"{'sum of medians':62s}        {tot:7.2f} ms per new 5 s prompt")
