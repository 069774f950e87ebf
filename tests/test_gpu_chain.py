"""The HIP path CHAINED (ids -> latent -> s2mel -> BigVGAN -> int16, every stage consuming the device output of the stage
before it) against the reference's own chain (`tests/golden/chain_tiny.npz`, make_golden.py `gen_chain`): north_star's
"bit-exact token ids under greedy decode and within 1e-3 max-abs on the fp32 waveform", end to end."""
import numpy as np
import pytest
import torch

from test_chain_golden import chain_cfgs, pcm_mismatch

pytestmark = pytest.mark.gpu


def test_hip_chain_vs_reference_chain(golden):
    import voice_tts_amd.s2mel as S2
    import voice_tts_amd.weights as WR
    from voice_tts_amd.pipeline import HotPath

    g = golden("chain_tiny.npz")
    dev = torch.device("cuda:0")
    gcfg, scfg, bcfg = chain_cfgs()
    sg, ss, sb = (int(x) for x in g["seeds"])
    n = len(g["ids"])
    hp = HotPath(gcfg, bcfg, dtype="f32", device=dev, max_batch=1, max_seq=128, max_frames=128)
    hp.load(WR.make_gpt_weights(gcfg, seed=sg, head_scale=50.0), WR.make_bigvgan_weights(bcfg, seed=sb))
    hp.attach_s2mel(S2.make_s2mel_weights(scfg, seed=ss), scfg)
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    cl = hp.conds_latent(t("cond32"), t("emo_vec"))
    text = torch.from_numpy(g["text"])
    emb, pad, P = hp.prepare_gpt_inputs(cl, text)
    ids = hp.generate([(emb, pad)], n, repetition_penalty=10.0)[0]
    assert ids.tolist() == g["ids"].tolist()  # bit-exact greedy ids (min reference margin 0.23)
    lat = hp.latent(cl, text, ids)  # device tensor, straight into the next stage
    e_lat = (lat.cpu() - torch.from_numpy(g["latent"])).abs().max().item()
    mel = hp.s2mel(lat, ids, t("prompt_condition"), t("ref_mel"), t("style"), n_timesteps=int(g["n_steps"]), noise=t("noise"))
    e_mel = (mel.cpu() - torch.from_numpy(g["mel"])).abs().max().item()
    scaled = hp.vocode(mel).cpu()  # [1, T] fp32 scaled to +-32767 (infer_v2.py:735-740)
    ref = torch.from_numpy(g["wav"])
    e_wav = (scaled - ref).abs().max().item() / 32767
    print(f"chain: latent err {e_lat:.2e}, mel err {e_mel:.2e}, waveform err {e_wav:.2e} (fp32 full scale 1.0)")
    assert e_lat <= 1e-4 and e_mel <= 2e-4 * max(1.0, float(np.abs(g["mel"]).max())) and e_wav <= 1e-4  # north_star asks 1e-3; measured 1.3e-6
    bad, worst, covered = pcm_mismatch(scaled, g["pcm"], g["wav"])
    assert bad == 0 and worst <= 1 and covered > 0.5, (bad, worst, covered)
