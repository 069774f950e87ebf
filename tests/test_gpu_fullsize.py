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
    """Throughput mode at full depth, config-1 shape: first logits within 2e-2 of the fp32 oracle's, >= 21 of 24 free-running
    greedy tokens equal (the 1100-step bound is test_bench_shape_bf16_1100_steps_vs_oracle)."""
    import voice_tts_amd.weights as WR
    from voice_tts_amd.gpt_engine import GptEngine

    W, orc, conds, text, embeds, mask, ids, margins, logits = gpt_full
    eng = GptEngine(WR.GPT_CFG, dtype="bf16", max_seq=256, max_batch=1, device=dev).load_state_dict(W)
    eng.prefill(0, embeds, 0)
    l0 = eng.read_logits(0)
    rel = np.abs(l0 - logits[0].numpy()).max() / np.abs(logits[0].numpy()).max()
    assert rel <= 2e-2, rel
    eng.decode(1, len(ids), repetition_penalty=10.0, suppress_stop=True)
    out, _ = eng.read(0)
    agree = int(np.sum(np.array(out.tolist()) == np.array(ids)))
    print(f"bf16 vs fp32-oracle: first-logit rel err {rel:.2e}, token agreement {agree}/{len(ids)}")
    assert agree >= 21, (agree, out.tolist(), ids)


# ---------------------------------------------------------------------------------------------------------------------------------
# The shape bench.py times (BASELINE configs[1]): 24 layers x D=1280, two sequences decoded together for 1100 steps from a
# 137-row prompt -- context 137 -> 1237, four 256-key attention buckets crossed, 8-step graphs.  One slot is a 117-row prompt
# with 3 left-padding rows, so lengths are unequal and the padding mask is live at full depth.
N_BENCH = 1100


@pytest.fixture(scope="module")
def bench_prompts(gpt_full):
    W, orc = gpt_full[0], gpt_full[1]
    g = torch.Generator().manual_seed(100)
    out = []
    for text in (torch.randint(2, 12000, (100,), generator=g),
                 torch.cat((torch.tensor([0, 1, 0]), torch.randint(2, 12000, (77,), generator=g)))):
        conds = torch.randn(34, 1280, generator=g) * 0.5
        fake, embeds, mask = orc.prepare_gpt_inputs(conds, text)
        out.append((embeds, mask, int((mask == 0).sum())))
    assert [len(m) for _, m, _ in out] == [137, 117] and [p for _, _, p in out] == [0, 3]
    return out


def _run_bench_shape(W, prompts, dtype, dev):
    """Free-running greedy decode exactly as bench.py issues it (B=2, repetition penalty 10, stop suppressed), cut into
    chunks so the logits can be read at: step 1, 2, around every 256-key bucket boundary of either slot, and step 1100."""
    import voice_tts_amd.weights as WR
    from voice_tts_amd.gpt_engine import GptEngine

    eng = GptEngine(WR.GPT_CFG, dtype=dtype, max_seq=137 + N_BENCH + 64, max_batch=2, device=dev).load_state_dict(W)
    for b, (emb, mask, pad) in enumerate(prompts):
        eng.prefill(b, emb, pad)
    stops = {1, 2, 64, N_BENCH}
    for _, mask, _ in prompts:
        for m in range(1, 6):
            for d in (-2, -1, 0, 1):
                k = 256 * m - len(mask) + d
                if 1 <= k <= N_BENCH:
                    stops.add(k)
    got = {0: [eng.read_logits(b).copy() for b in range(2)]}
    done = 0
    for k in sorted(stops):
        eng.decode(2, k - done, repetition_penalty=10.0, suppress_stop=True)
        done = k
        got[k] = [eng.read_logits(b).copy() for b in range(2)]
    ids = [eng.read(b)[0][:N_BENCH] for b in range(2)]
    assert all(len(i) == N_BENCH for i in ids)
    return ids, got


def _oracle_rows(orc, prompts, ids):
    from oracle import gpt as OG

    torch.set_num_threads(max(torch.get_num_threads(), 8))
    out = []
    for (emb, mask, pad), i in zip(prompts, ids):
        rows = OG.teacher_forced_logits(orc, emb, mask, i.tolist())  # ONE causal pass over the 1237 (1217) rows
        picks, margins = OG.greedy_choices(rows, len(mask), i.tolist(), theta=10.0, suppress_stop=True)
        out.append((rows, picks, margins))
    return out


def test_bench_shape_fp32_1100_steps_vs_oracle(gpt_full, bench_prompts, dev):
    """Parity mode at the benchmarked shape: every one of the 2 x 1100 device tokens is the oracle's greedy choice given the
    same history (exact wherever the oracle's own top-2 margin exceeds 1e-4 of the logit scale), and the logits read at 30+
    steps -- including both sides of every 256-key bucket switch -- are within 3e-4 of the logit scale."""
    W, orc = gpt_full[0], gpt_full[1]
    ids, got = _run_bench_shape(W, bench_prompts, "f32", dev)
    ref = _oracle_rows(orc, bench_prompts, ids)
    for b in range(2):
        rows, picks, margins = ref[b]
        scale = float(rows.abs().max())
        worst = max(np.abs(got[k][b] - rows[k].numpy()).max() for k in got) / scale
        # "near tie" = the oracle's own top-2 margin below 1e-4 of the logit scale (the measured logit error is ~2e-6 of it)
        close = [k for k in range(N_BENCH) if margins[k] < 1e-4 * scale]
        differ = [k for k in range(N_BENCH) if picks[k] != int(ids[b][k])]
        wrong = [k for k in differ if margins[k] >= 1e-4 * scale]
        print(f"fp32 slot {b}: logits rel err {worst:.2e} over {len(got)} read points, {len(close)} near-tie steps, "
              f"{len(differ)} differing tokens, {len(wrong)} of them outside near-ties")
        assert worst <= 3e-4, (b, worst)
        assert not wrong, (b, wrong[:5])
        assert len(close) <= 10 and len(differ) <= len(close)


def test_bench_shape_bf16_1100_steps_vs_oracle(gpt_full, bench_prompts, dev):
    """The benchmarked mode (bf16 weights + KV, fp32 accumulate) against the fp32 CPU oracle, teacher-forced on the device's own
    ids: stated bounds -- logits within 1.5e-2 of the logit scale at every read point (measured 5.6e-3), and the device token
    equals the fp32 oracle's greedy choice at >= 95 % of the 2 x 1100 steps (measured 97.8 % / 98.5 %)."""
    W, orc = gpt_full[0], gpt_full[1]
    ids, got = _run_bench_shape(W, bench_prompts, "bf16", dev)
    ref = _oracle_rows(orc, bench_prompts, ids)
    for b in range(2):
        rows, picks, margins = ref[b]
        scale = float(rows.abs().max())
        errs = {k: np.abs(got[k][b] - rows[k].numpy()).max() / scale for k in got}
        agree = sum(int(picks[k] == int(ids[b][k])) for k in range(N_BENCH)) / N_BENCH
        top1 = sum(int(np.argmax(got[k][b]) == int(rows[k].argmax())) for k in got) / len(got)
        print(f"bf16 slot {b}: logits rel err max {max(errs.values()):.2e} (step {max(errs, key=errs.get)}), greedy agreement {agree:.4f}, "
              f"raw top-1 agreement at read points {top1:.3f}")
        assert max(errs.values()) <= 1.5e-2, (b, errs)
        assert agree >= 0.95, (b, agree)


def test_bigvgan_full_size_config5_mel(dev):
    """BASELINE configs[4]: the 1000-frame microbench mel through the production generator vs the CPU oracle, <= 1e-3
    (north_star) -- asserted at 5e-5 (measured 6e-6)."""
    from oracle import vocoder as OV
    import voice_tts_amd.weights as WR
    from voice_tts_amd.bigvgan import BigVGAN

    W = WR.make_bigvgan_weights(WR.BIGVGAN_CFG, seed=1234)
    mel = (torch.randn(1, 80, 1000, generator=torch.Generator().manual_seed(6)) * 2 - 4).clamp(-11.5, 2)
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    ref = OV.bigvgan_forward(mel, W)
    m = BigVGAN(WR.BIGVGAN_CFG, max_frames=1024, device=dev).load_state_dict(W)
    wav = m(mel.to(dev)).cpu()
    assert wav.shape == ref.shape == (1, 1, 256000)
    err = (wav - ref).abs().max().item()
    print(f"BigVGAN F=1000: max|err| {err:.2e}, ref max {ref.abs().max().item():.3f}")
    assert ref.abs().max() > 0.05 and err <= 5e-5, err


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
