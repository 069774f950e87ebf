"""Full production shapes (BASELINE.json configs): parity against the CPU oracle on bounded samples."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def gpt_full():
    import voice_tts_amd.weights as WR
    from oracle import gpt as OG

    W = WR.make_gpt_weights(WR.GPT_CFG, seed=1234)
    orc = OG.GptOracle(W, WR.GPT_CFG["layers"], WR.GPT_CFG["heads"])
    g = torch.Generator().manual_seed(2)
    conds = torch.randn(34, 1280, generator=g) * 0.5
    text = torch.randint(2, 12000, (20,), generator=g)  # config 1: 20-token text -> P = 57
    fake, embeds, mask = orc.prepare_gpt_inputs(conds, text)
    n = 24
    ids, margins, logits = OG.generate_greedy(orc, embeds, mask, n, return_logits=True, suppress_stop=True)
    return W, orc, conds, text, embeds, mask, ids, margins, logits


def test_gpt_24_layers_fp32_greedy_bit_exact(gpt_full, dev):
    """Config 1 shapes (20-token text, P=57), all 24 layers at D=1280: greedy ids equal the oracle's."""
    import voice_tts_amd.weights as WR
    from voice_tts_amd.gpt_engine import GptEngine

    W, orc, conds, text, embeds, mask, ids, margins, logits = gpt_full
    eng = GptEngine(WR.GPT_CFG, dtype="f32", max_seq=256, max_batch=1, device=dev).load_state_dict(W)
    eng.prefill(0, embeds, 0)
    l0 = eng.read_logits(0)
    assert np.abs(l0 - logits[0].numpy()).max() <= 3e-4 * np.abs(logits[0].numpy()).max()
    eng.decode(1, len(ids), repetition_penalty=10.0, suppress_stop=True)
    out, _ = eng.read(0)
    # bit-exact wherever the oracle's own top-2 margin is above fp32 reduction noise (1e-3 of the logit scale)
    scale = float(logits.abs().max())
    first_close = next((i for i, m in enumerate(margins) if m < 1e-3 * scale), len(ids))
    assert out.tolist()[:first_close] == ids[:first_close]
    assert first_close >= len(ids) - 1 or out.tolist() == ids, (first_close, margins)
    # latent pass at full depth
    codes = torch.tensor(ids[:16])
    t = torch.cat((torch.tensor([0]), text, torch.tensor([1])))
    prefix = torch.cat((conds, W["text_embedding.weight"][t] + W["text_pos_embedding.emb.weight"][: t.numel()]), 0)
    lat = eng.latent(prefix, codes).cpu()
    ref = orc.latent_pass(conds, text, codes)
    assert (lat - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())


def test_gpt_24_layers_bf16_token_agreement(gpt_full, dev):
    """Throughput mode at full depth: reports agreement with the fp32 oracle (not required to be exact)."""
    import voice_tts_amd.weights as WR
    from voice_tts_amd.gpt_engine import GptEngine

    W, orc, conds, text, embeds, mask, ids, margins, logits = gpt_full
    eng = GptEngine(WR.GPT_CFG, dtype="bf16", max_seq=256, max_batch=1, device=dev).load_state_dict(W)
    eng.prefill(0, embeds, 0)
    l0 = eng.read_logits(0)
    rel = np.abs(l0 - logits[0].numpy()).max() / np.abs(logits[0].numpy()).max()
    assert rel <= 5e-2, rel
    eng.decode(1, len(ids), repetition_penalty=10.0, suppress_stop=True)
    out, _ = eng.read(0)
    agree = int(np.sum(np.array(out.tolist()) == np.array(ids)))
    print(f"bf16 vs fp32-oracle: first-logit rel err {rel:.2e}, token agreement {agree}/{len(ids)}")
    assert out.tolist()[0] == ids[0] or margins[0] < 0.05 * float(logits.abs().max())


@pytest.mark.parametrize("B", [1, 2, 3])
def test_fused_mlp_kernel_matches_the_split_kernels(gpt_full, dev, monkeypatch, B):
    """bf16 decode at full size: the opt-in fused MLP launch (IXTTS_MLP=fused: LN2 + c_fc + gelu + c_proj with the hand-off
    inside each XCD, partial sums completed by the next layer's QKV kernel) against the default split FC / MLP-out launches.
    The two sum c_proj in different orders, so logits agree to rounding (not bit for bit) and greedy tokens are the same; the
    hand-off must never time out (a time-out makes `read` raise)."""
    import voice_tts_amd.weights as WR
    from voice_tts_amd.gpt_engine import GptEngine

    W, orc, conds, text, embeds, mask, ids, margins, logits = gpt_full
    n = 40

    def run(split):
        if split:
            monkeypatch.delenv("IXTTS_MLP", raising=False)
        else:
            monkeypatch.setenv("IXTTS_MLP", "fused")
        eng = GptEngine(WR.GPT_CFG, dtype="bf16", max_seq=256, max_batch=B, device=dev).load_state_dict(W)
        for b in range(B):
            eng.prefill(b, embeds[: embeds.shape[0] - 3 * b], 0)  # different lengths per slot
        eng.decode(B, 1, repetition_penalty=10.0, suppress_stop=True)
        first = [eng.read_logits(b).copy() for b in range(B)]
        eng.decode(B, n - 1, repetition_penalty=10.0, suppress_stop=True)
        return first, [eng.read(b)[0][:n].tolist() for b in range(B)], [eng.read_logits(b).copy() for b in range(B)]

    f1, t1, l1 = run(False)
    f2, t2, l2 = run(True)
    scale = max(float(np.abs(x).max()) for x in f2)
    for b in range(B):
        assert np.abs(f1[b] - f2[b]).max() <= 2e-3 * scale, (b, np.abs(f1[b] - f2[b]).max(), scale)
        assert t1[b] == t2[b], (b, [i for i, (x, y) in enumerate(zip(t1[b], t2[b])) if x != y][:3])
        assert np.abs(l1[b] - l2[b]).max() <= 5e-3 * scale


def test_bigvgan_full_size_waveform(dev):
    """Production generator (1536 channels, 112 M params): waveform within 1e-3 max-abs of the CPU oracle
    (north_star); asserted at 2e-4."""
    from oracle import vocoder as OV
    import voice_tts_amd.weights as WR
    from voice_tts_amd.bigvgan import BigVGAN

    W = WR.make_bigvgan_weights(WR.BIGVGAN_CFG, seed=1234)
    mel = (torch.randn(1, 80, 24, generator=torch.Generator().manual_seed(6)) * 2 - 4).clamp(-11.5, 2)
    ref = OV.bigvgan_forward(mel, W)
    m = BigVGAN(WR.BIGVGAN_CFG, max_frames=32, device=dev).load_state_dict(W)
    wav = m(mel.to(dev)).cpu()
    assert wav.shape == ref.shape == (1, 1, 24 * 256)
    err = (wav - ref).abs().max().item()
    assert ref.abs().max() > 0.05 and err <= 2e-4, err
    # PCM conversion of the pipeline: truncation toward zero after the clamp (infer_v2.py:740,772)
    pcm = torch.clamp(32767 * wav, -32767.0, 32767.0).to(torch.int16)
    assert (pcm.int() - OV.pcm16(ref).int()).abs().max().item() <= 8
