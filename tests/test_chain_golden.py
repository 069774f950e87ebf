"""SURVEY 8(d) config 1 on twins: the CHAIN ids -> latent -> s2mel (injected CFM noise) -> BigVGAN -> int16,
each stage fed by the previous one's own output, against `tests/golden/chain_tiny.npz` -- produced by the reference's own
UnifiedVoice / MyModel / CFM / BigVGAN classes chained as infer_v2.py:641-744 chains them (make_golden.py `gen_chain`).
This file holds the CPU side (oracle/ + the torch glue on host tensors) to that fixture; tests/test_gpu_chain.py holds the
HIP path to it."""
import numpy as np
import torch

import voice_tts_amd.s2mel as S2
import voice_tts_amd.weights as WR
from oracle import gpt as OG
from oracle import vocoder as OV


def chain_cfgs():
    gcfg = WR.tiny_gpt_cfg(model_dim=1280, layers=2, heads=20)
    scfg = S2.tiny_s2mel_cfg(gpt_dim=1280, semantic_dim=1024, lr_in_channels=1024, codebook_size=8194, hidden_dim=128, num_heads=2,
                             wavenet_hidden=128, depth=3)
    return gcfg, scfg, WR.tiny_bigvgan_cfg(64)


def pcm_mismatch(wav_scaled, pcm_ref, wav_ref):
    """int16 truncation (infer_v2.py:781) is discontinuous at integers: samples whose reference value sits within `tol` of
    an integer step may legitimately land one LSB away; everywhere else the PCM must be identical."""
    got = wav_scaled.to(torch.int16).int()
    ref = torch.as_tensor(pcm_ref).int()
    frac = (torch.as_tensor(wav_ref) - torch.as_tensor(wav_ref).round()).abs()
    tol = float((wav_scaled - torch.as_tensor(wav_ref)).abs().max()) + 1e-6
    away = frac > tol
    return int(((got != ref) & away).sum()), int((got - ref).abs().max()), float(away.float().mean())


def test_cpu_chain_vs_reference_chain(golden):
    g = golden("chain_tiny.npz")
    gcfg, scfg, bcfg = chain_cfgs()
    sg, ss, sb = (int(x) for x in g["seeds"])
    Wg = WR.make_gpt_weights(gcfg, seed=sg, head_scale=50.0)
    orc = OG.GptOracle(Wg, gcfg["layers"], gcfg["heads"])
    cl = orc.conds_latent(torch.from_numpy(g["cond32"]), torch.from_numpy(g["emo_vec"]))
    assert torch.allclose(cl, torch.from_numpy(g["conds_latent"]), atol=1e-6)
    text = torch.from_numpy(g["text"])
    fake, emb, mask = orc.prepare_gpt_inputs(cl, text)
    n = len(g["ids"])
    ids, margins = OG.generate_greedy(orc, emb, mask, n)
    assert ids == g["ids"].tolist()
    lat = orc.latent_pass(cl, text, ids)
    assert (lat - torch.from_numpy(g["latent"])).abs().max().item() <= 1e-4
    m = S2.S2Mel(S2.make_s2mel_weights(scfg, seed=ss), scfg, device="cpu")
    t = lambda k: torch.from_numpy(g[k])
    mel = m(lat.unsqueeze(0), torch.tensor([ids]), torch.tensor([n]), t("prompt_condition"), t("ref_mel"), t("style"),
            n_timesteps=int(g["n_steps"]), inference_cfg_rate=0.7, noise=t("noise"))
    assert (mel - t("mel")).abs().max().item() <= 1e-4 * max(1.0, float(np.abs(g["mel"]).max()))
    wav = OV.bigvgan_forward(mel, WR.make_bigvgan_weights(bcfg, seed=sb), bcfg)
    scaled = torch.clamp(32767 * wav.squeeze().unsqueeze(0), -32767.0, 32767.0)
    assert (scaled - t("wav")).abs().max().item() / 32767 <= 1e-3  # north_star: 1e-3 max-abs on the fp32 waveform
    bad, worst, covered = pcm_mismatch(scaled, g["pcm"], g["wav"])
    assert bad == 0 and worst <= 1 and covered > 0.5, (bad, worst, covered)
