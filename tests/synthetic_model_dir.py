"""Test helper: write a complete, tiny, seeded `model_dir` in the layout `IndexTTS2.__init__` reads -- config.yaml,
gpt.pth, s2mel.pth, bigvgan_generator.pt, semantic_codec/model.safetensors, campplus_cn_common.bin, w2v-bert-2.0/ (a randomly
initialised `Wav2Vec2BertModel` saved with `save_pretrained`), wav2vec2bert_stats.pt, feat1.pt / feat2.pt, bpe.model -- so the
drop-in constructor is exercised through its own file loaders.  Everything is synthetic; nothing is downloaded."""
import io
import os
import shutil
import wave

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
HID = 32  # w2v-bert hidden size of the twin = conditioning input_size = codec hidden_size = s2mel semantic_dim


def twin_cfgs():
    import voice_tts_amd.conditioning as CD
    import voice_tts_amd.s2mel as S2
    import voice_tts_amd.weights as WR

    gcfg = WR.tiny_gpt_cfg(model_dim=1280, layers=2, heads=20, number_text_tokens=400)  # MyModel.gpt_layer is fixed 1280 -> ... -> 1024
    ccfg = CD.tiny_cond_cfg(model_dim=1280, input_size=HID, emo_dim=40, cond_num=32)
    scfg = S2.tiny_s2mel_cfg(gpt_dim=1280, semantic_dim=HID, lr_in_channels=HID, codebook_size=8194, style_dim=192, hidden_dim=128, num_heads=2,
                             wavenet_hidden=128, depth=3)
    qcfg = dict(codebook_size=8194, hidden_size=HID, codebook_dim=8, vocos_dim=24, vocos_intermediate_dim=48, vocos_num_layers=2)
    return gcfg, ccfg, scfg, qcfg, WR.tiny_bigvgan_cfg(64)


def write_model_dir(root, seed=300):
    import voice_tts_amd.conditioning as CD
    import voice_tts_amd.prompt as PR
    import voice_tts_amd.s2mel as S2
    import voice_tts_amd.weights as WR
    from safetensors.torch import save_file
    from transformers import SeamlessM4TFeatureExtractor, Wav2Vec2BertConfig, Wav2Vec2BertModel

    os.makedirs(root, exist_ok=True)
    gcfg, ccfg, scfg, qcfg, bcfg = twin_cfgs()
    Wg = WR.make_gpt_weights(gcfg, seed=seed, head_scale=50.0)
    Wg.update(CD.make_cond_weights(ccfg, seed=seed + 1))
    torch.save({"model": Wg}, os.path.join(root, "gpt.pth"))
    Ws = S2.make_s2mel_weights(scfg, seed=seed + 2)
    net = {}
    for k, v in Ws.items():
        mod, rest = k.split(".", 1)
        if mod != "quantizer":  # vq2emb's tensors live in the codec checkpoint (infer_v2.py:714)
            net.setdefault(mod, {})["module." + rest if mod == "cfm" else rest] = v
    torch.save({"net": net, "epoch": 0}, os.path.join(root, "s2mel.pth"))
    torch.save({"generator": WR.make_bigvgan_weights(bcfg, seed=seed + 3)}, os.path.join(root, "bigvgan_generator.pt"))
    os.makedirs(os.path.join(root, "semantic_codec"), exist_ok=True)
    save_file({k: v.contiguous() for k, v in PR.make_codec_weights(qcfg, seed=seed + 4).items()}, os.path.join(root, "semantic_codec", "model.safetensors"))
    Wc = PR.make_camplus_weights(seed=seed + 5)
    Wc["xvector.tdnn.nonlinear.batchnorm.num_batches_tracked"] = torch.tensor(7)  # integer buffers of the real file are ignored
    torch.save(Wc, os.path.join(root, "campplus_cn_common.bin"))
    torch.manual_seed(seed + 6)
    w2v = Wav2Vec2BertModel(Wav2Vec2BertConfig(hidden_size=HID, num_hidden_layers=18, num_attention_heads=2, intermediate_size=48,
                                               feature_projection_input_dim=160, conv_depthwise_kernel_size=7, add_adapter=False)).eval()
    w2v.save_pretrained(os.path.join(root, "w2v-bert-2.0"))
    SeamlessM4TFeatureExtractor(feature_size=80, num_mel_bins=80, sampling_rate=16000, stride=2, padding_value=1).save_pretrained(os.path.join(root, "w2v-bert-2.0"))
    g = torch.Generator().manual_seed(seed + 7)
    torch.save({"mean": 0.1 * torch.randn(HID, generator=g), "var": 0.5 + torch.rand(HID, generator=g)}, os.path.join(root, "wav2vec2bert_stats.pt"))
    emo_num = [3, 4, 2, 3, 2, 2, 3, 5]
    torch.save(torch.randn(sum(emo_num), 192, generator=g), os.path.join(root, "feat1.pt"))
    torch.save(0.1 * torch.randn(sum(emo_num), gcfg["model_dim"], generator=g), os.path.join(root, "feat2.pt"))
    shutil.copy(os.path.join(HERE, "golden", "tiny_bpe.model"), os.path.join(root, "bpe.model"))
    cfg = {
        "gpt": {**{k: gcfg[k] for k in gcfg}, "condition_type": "conformer_perceiver",
                "condition_module": {**ccfg["condition_module"], "input_layer": "conv2d2"}, "emo_condition_module": {**ccfg["emo_condition_module"], "input_layer": "conv2d2"}},
        "semantic_codec": qcfg,
        "s2mel": {"preprocess_params": {"sr": 22050, "spect_params": {"n_fft": 1024, "win_length": 1024, "hop_length": 256, "n_mels": 80, "fmin": 0, "fmax": "None"}},
                  "style_encoder": {"dim": 192},
                  "length_regulator": {"channels": scfg["lr_channels"], "in_channels": HID, "sampling_ratios": [1] * scfg["lr_n_blocks"]},
                  "DiT": {"hidden_dim": 128, "num_heads": 2, "depth": 3, "in_channels": 80, "content_dim": scfg["content_dim"]},
                  "wavenet": {"hidden_dim": 128, "num_layers": scfg["wavenet_layers"], "kernel_size": scfg["wavenet_kernel"], "dilation_rate": 1}},
        "gpt_checkpoint": "gpt.pth", "w2v_stat": "wav2vec2bert_stats.pt", "s2mel_checkpoint": "s2mel.pth", "emo_matrix": "feat2.pt", "spk_matrix": "feat1.pt",
        "emo_num": emo_num, "qwen_emo_path": "qwen0.6bemo4-merge/", "vocoder": {"type": "bigvgan", "name": "bigvgan_generator.pt"},
        "dataset": {"bpe_model": "bpe.model"}, "version": 2.0,
    }
    with open(os.path.join(root, "config.yaml"), "w") as f:
        yaml.safe_dump(cfg, f)
    return os.path.join(root, "config.yaml"), cfg


def synthetic_wav_bytes(seconds=1.5, sr=24000, seed=0):
    """A mono 16-bit WAVE stream: a few decaying harmonics + noise (stands in for a speaker prompt recording)."""
    rng = np.random.RandomState(seed)
    t = np.arange(int(seconds * sr)) / sr
    x = sum(a * np.sin(2 * np.pi * f * t + p) for a, f, p in ((0.3, 140, 0), (0.2, 280, 1), (0.1, 420, 2), (0.05, 1900, 3)))
    x = (x * (0.6 + 0.4 * np.sin(2 * np.pi * 3 * t)) + 0.02 * rng.randn(t.size)).astype(np.float32)
    b = io.BytesIO()
    with wave.open(b, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(sr)
        f.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())
    return b.getvalue()
