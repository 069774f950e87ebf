"""SURVEY.md 8(d) "Config 1" at PRODUCTION width, every stage consuming the device output of the stage before it, against
the CPU path run on the same seeded inputs: conditioning encoders (6 / 4 conformer blocks, 512 wide) -> conds_latent -> 24-layer
x 1280 GPT, 220 greedy codes from a 57-row prompt (fp32 engine) -> latent forward -> s2mel at S2MEL_CFG (hidden 512, 13 DiT
layers, 25 Euler steps, CFG 0.7, injected CFM noise of seed 3, 258-frame prompt + 378 frames) -> BigVGAN (112 M parameters)
on 378 frames -> int16 PCM.  north_star: token ids bit-exact under greedy decode, waveform within 1e-3 max-abs in fp32.

CPU side: `oracle/` for the GPT and the vocoder (pinned to the reference's classes, tests/test_oracle_golden.py); the s2mel and
conditioning glue are torch restatements pinned to fixtures of the reference's own classes on CPU
(tests/test_s2mel_golden.py, tests/test_conditioning_golden.py) -- the same code on host tensors is their CPU leg here."""
import numpy as np
import pytest
import torch

from test_chain_golden import pcm_mismatch

pytestmark = pytest.mark.gpu

N_CODES, T_REF = 220, 258


def _inputs():
    text = torch.randint(2, 12000, (20,), generator=torch.Generator().manual_seed(1))  # 20 tokens -> P = 34 + 22 + 1 = 57
    g2 = torch.Generator().manual_seed(2)  # 3 s speaker prompt: 149 w2v-bert frames, 258 mel frames
    spk = torch.randn(1, 149, 1024, generator=g2)
    ref_mel = torch.randn(1, 80, T_REF, generator=g2) * 2 - 4
    style = torch.randn(1, 192, generator=g2)
    prompt_condition = torch.randn(1, T_REF, 512, generator=g2)
    frames = int(N_CODES * 1.72)
    noise = torch.randn(1, 80, T_REF + frames, generator=torch.Generator().manual_seed(3))
    return text, spk, ref_mel, style, prompt_condition, noise, frames


def test_config1_chain_at_production_width_vs_cpu_path():
    import voice_tts_amd.conditioning as CD
    import voice_tts_amd.s2mel as S2
    import voice_tts_amd.weights as WR
    from oracle import gpt as OG
    from oracle import vocoder as OV
    from voice_tts_amd.pipeline import HotPath

    import time

    t_ = [time.perf_counter()]

    def lap(what):
        t_.append(time.perf_counter())
        print(f"[chain timing] {what}: {t_[-1] - t_[-2]:.1f} s", flush=True)

    dev = torch.device("cuda:0")
    Wg = WR.make_gpt_weights(WR.GPT_CFG, seed=1234)
    Wc = CD.make_cond_weights(CD.COND_CFG, seed=1234)
    Ws = S2.make_s2mel_weights(S2.S2MEL_CFG, seed=1234)
    Wb = WR.make_bigvgan_weights(WR.BIGVGAN_CFG, seed=1234)
    lap("weights")
    text, spk, ref_mel, style, pc, noise, frames = _inputs()
    assert frames == 378
    rescale = lambda cl: cl * (0.5 / cl.std().clamp_min(1e-6))  # synthetic encoder weights: keep the GPT prefix at the scale it is built for

    # ---- CPU path
    cond32, emovec = CD.Conditioning(Wc, CD.COND_CFG, device="cpu").encode_prompt(spk, None, 1.0)
    orc = OG.GptOracle(Wg, WR.GPT_CFG["layers"], WR.GPT_CFG["heads"])
    cl_ref = rescale(orc.conds_latent(cond32, emovec))
    lap("cpu conditioning")
    fake, emb_ref, mask = orc.prepare_gpt_inputs(cl_ref, text)
    assert len(mask) == 57
    ids_ref, margins, logits = OG.generate_greedy(orc, emb_ref, mask, N_CODES, return_logits=True, suppress_stop=True)
    scale = float(logits.abs().max())
    lap("cpu greedy decode")
    lat_ref = orc.latent_pass(cl_ref, text, torch.tensor(ids_ref))
    mel_ref = S2.S2Mel(Ws, S2.S2MEL_CFG, device="cpu")(lat_ref.unsqueeze(0), torch.tensor(ids_ref).view(1, -1), torch.tensor([N_CODES]), pc, ref_mel, style,
                                                        n_timesteps=25, inference_cfg_rate=0.7, noise=noise)
    lap("cpu latent + s2mel")
    mel_ref_c = mel_ref.clamp(-11.5, 2.0)  # (synthetic s2mel weights: keep the log-mel range the vocoder is built for -- both sides alike)
    wav_ref = OV.bigvgan_forward(mel_ref_c, Wb)
    scaled_ref = torch.clamp(32767 * wav_ref.squeeze(1), -32767.0, 32767.0)
    pcm_ref = OV.pcm16(wav_ref.squeeze(1))
    lap("cpu bigvgan")

    # ---- HIP path, chained on the device
    hp = HotPath(dtype="f32", device=dev, max_batch=1, max_seq=57 + N_CODES + 64, max_frames=frames).load(Wg, Wb)
    hp.attach_s2mel(Ws).attach_conditioning(Wc)
    cl = rescale(hp.conds_from_prompt(spk.to(dev)))
    e_cl = (cl.cpu() - cl_ref).abs().max().item()
    emb, pad, P = hp.prepare_gpt_inputs(cl, text)
    ids = hp.generate([(emb, pad)], N_CODES, repetition_penalty=10.0, fixed_length=True)[0]
    differ = [k for k in range(N_CODES) if int(ids[k]) != ids_ref[k]]
    lat = hp.latent(cl, text, ids)
    e_lat = (lat.cpu() - lat_ref).abs().max().item()
    mel = hp.s2mel(lat, ids, pc.to(dev), ref_mel.to(dev), style.to(dev), n_timesteps=25, noise=noise.to(dev))
    e_mel = (mel.cpu() - mel_ref).abs().max().item()
    scaled = hp.vocode(mel.clamp(-11.5, 2.0)).cpu()
    e_wav = (scaled - scaled_ref).abs().max().item() / 32767
    lap("device chain (load + run)")
    print(f"config-1 chain at production width: conds_latent err {e_cl:.2e}, {len(differ)} of {N_CODES} ids differ (min oracle margin "
          f"{min(margins):.3f} of logit scale {scale:.1f}), latent err {e_lat:.2e} (max {lat_ref.abs().max().item():.2f}), mel err {e_mel:.2e} "
          f"(max {mel_ref.abs().max().item():.2f}), waveform err {e_wav:.2e} of full scale (peak {wav_ref.abs().max().item():.3f})")
    assert e_cl <= 1e-4 * max(1.0, cl_ref.abs().max().item())
    assert not differ, differ[:5]
    assert e_lat <= 1e-4 * max(1.0, lat_ref.abs().max().item())
    assert e_mel <= 1e-4 * max(1.0, mel_ref.abs().max().item())
    assert wav_ref.abs().max().item() > 0.05 and e_wav <= 2e-4  # north_star: 1e-3
    # int16 PCM (truncation toward zero, infer_v2.py:781): never further off than the waveform error rounded up + 1 LSB; and where the
    # error is well under an LSB, identical except where the reference value sits within the error of an integer step
    lsb = e_wav * 32767
    assert int((scaled.to(torch.int16).int() - pcm_ref.int()).abs().max()) <= int(lsb) + 1
    if lsb < 0.4:
        bad, worst, covered = pcm_mismatch(scaled, pcm_ref, scaled_ref)
        assert bad == 0 and worst <= 1 and covered > 0.1, (bad, worst, covered)
