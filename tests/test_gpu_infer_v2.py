"""Outer boundary: `indextts.infer_v2.IndexTTS2.infer(...)` contract (infer_v2.py:438-461,740-783) with the two
hot stages on the HIP path and synthetic stand-ins for the PyTorch glue stages this repo does not build."""
import os
import sys
import wave

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


class FakeGlue:
    """Deterministic stand-in for the glue stages (shapes as in SURVEY.md 8(d) config 1)."""

    def __init__(self, D, dev, frames_per_code=1.72):
        self.D, self.dev = D, dev
        self.calls = dict(speaker=0, emotion=0, s2mel=0)
        self.g = torch.Generator().manual_seed(5)

    def tokenize(self, text, max_text_tokens_per_segment, quick_streaming_tokens=0):
        ids = [2 + (ord(c) % 190) for c in text]
        return [ids[i:i + max_text_tokens_per_segment] for i in range(0, len(ids), max_text_tokens_per_segment)]

    def speaker(self, spk_audio_prompt):
        self.calls["speaker"] += 1
        return dict(spk_cond_emb=torch.randn(1, 20, 16, generator=self.g).to(self.dev), style=torch.randn(1, 192, generator=self.g).to(self.dev),
                    prompt_condition=None, ref_mel=None)

    def emotion(self, emo_audio_prompt):
        self.calls["emotion"] += 1
        return torch.randn(1, 20, 16, generator=self.g).to(self.dev)

    def emo_vector_mix(self, emo_vector, style, use_random):
        w = torch.tensor(emo_vector)
        return torch.full((1, self.D), 0.01 * float(w.sum()), device=self.dev), float(w.sum())

    def merge_emovec(self, spk_cond_emb, emo_cond_emb, alpha):
        base = spk_cond_emb.mean(dim=(1, 2)).reshape(1, 1).expand(1, self.D)
        emo = emo_cond_emb.mean(dim=(1, 2)).reshape(1, 1).expand(1, self.D)
        return (base + alpha * (emo - base)).contiguous()

    def get_conditioning(self, spk_cond_emb):
        return torch.linspace(-0.5, 0.5, 32 * self.D, device=self.dev).reshape(32, self.D)

    def s2mel(self, latent, codes, code_lens, speaker):
        self.calls["s2mel"] += 1
        F = int(int(code_lens[0]) * 1.72)
        assert latent.shape == (1, codes.shape[1], self.D)
        mel = (torch.randn(1, 80, F, generator=torch.Generator().manual_seed(F)) * 2 - 4).clamp(-11.5, 2)
        return mel.to(self.dev)


@pytest.fixture(scope="module")
def tts():
    import voice_tts_amd.weights as WR
    from indextts.infer_v2 import IndexTTS2

    dev = torch.device("cuda:0")
    gcfg = WR.tiny_gpt_cfg(model_dim=128, layers=2, heads=2)
    bcfg = WR.tiny_bigvgan_cfg(64)
    Wg = WR.make_gpt_weights(gcfg, seed=7)
    Wb = WR.make_bigvgan_weights(bcfg, seed=8)
    glue = FakeGlue(128, dev)
    m = IndexTTS2(cfg_path=None, model_dir="/nonexistent", use_fp16=False, device="cuda:0", use_cuda_kernel=True, glue=glue,
                  gpt_state_dict=Wg, bigvgan_state_dict=Wb, gpt_cfg=gcfg, bigvgan_cfg=bcfg, max_seq=192, max_frames=128)
    return m, glue


def test_infer_returns_reference_shaped_audio(tts, tmp_path):
    m, glue = tts
    text = "abcdefghijklmnopqrstuvwxyz0123456789" * 2  # 72 tokens -> 3 segments of <= 30
    # greedy path through the served kwargs (top_k=1), short generations
    res = m.infer("spk.wav", text, None, max_text_tokens_per_segment=30, num_beams=1, top_k=1, max_mel_tokens=20)
    sr, pcm = res
    assert sr == 22050 and pcm.dtype == np.int16 and pcm.ndim == 2 and pcm.shape[1] == 1
    frames = int(20 * 1.72)
    expect = 3 * frames * 256 + 2 * int(22050 * 0.2)  # 3 segments + 2 x 200 ms silence (infer_v2.py:283-304,752)
    assert pcm.shape[0] == expect
    assert np.abs(pcm).max() <= 32767
    assert glue.calls["speaker"] == 1 and glue.calls["s2mel"] == 3
    t = m.last_timing
    assert t["audio_length"] == pytest.approx(expect / 22050)
    # same speaker prompt object again -> cached (infer_v2.py:508-550); output file variant
    out = str(tmp_path / "o" / "x.wav")
    assert m.infer("spk.wav", "hello world", out, num_beams=1, top_k=1, max_mel_tokens=12) == out
    with wave.open(out) as f:
        assert (f.getnchannels(), f.getsampwidth(), f.getframerate()) == (1, 2, 22050)
        assert f.getnframes() == int(12 * 1.72) * 256
    # empty text -> no segments -> None (infer_v2.py:460-461)
    assert m.infer("spk.wav", "", None) is None


def test_stream_return_and_served_defaults(tts):
    m, glue = tts
    # served defaults: num_beams=3 beam-sample, top_k 30, top_p .8, temperature .8, repetition_penalty 10 (infer_v2.py:598-606)
    gen = m.infer("spk2.wav", "abcdefghij" * 5, None, max_text_tokens_per_segment=25, stream_return=True, max_mel_tokens=16,
                  emo_vector=[0.1, 0, 0, 0, 0, 0, 0, 0.2], seed=11)
    chunks = list(gen)
    # per segment: wav tensor then the silence tensor (infer_v2.py:745-749); nothing else when stream_return
    assert len(chunks) == 4
    for i, c in enumerate(chunks):
        assert isinstance(c, torch.Tensor) and c.dim() == 2 and c.shape[0] == 1 and c.device.type == "cpu"
        if i % 2 == 1:
            assert c.shape[1] == int(22050 * 0.2) and float(c.abs().max()) == 0.0
        else:
            assert c.shape[1] % 256 == 0 and c.shape[1] > 0 and float(c.abs().max()) <= 32767.0


def test_missing_glue_is_loud():
    import voice_tts_amd.weights as WR
    from indextts.infer_v2 import IndexTTS2

    gcfg = WR.tiny_gpt_cfg(model_dim=128, layers=2, heads=2)
    bcfg = WR.tiny_bigvgan_cfg(64)
    m = IndexTTS2(cfg_path=None, model_dir="/nonexistent", device="cuda:0", gpt_state_dict=WR.make_gpt_weights(gcfg, seed=7),
                  bigvgan_state_dict=WR.make_bigvgan_weights(bcfg, seed=8), gpt_cfg=gcfg, bigvgan_cfg=bcfg, max_seq=96, max_frames=32)
    assert m.device.type == "cuda" and m.use_fp16 is False  # attributes server.py:306-307 reads
    with pytest.raises(NotImplementedError):
        m.infer("spk.wav", "hi", None)
    with pytest.raises(FileNotFoundError):
        IndexTTS2(cfg_path=None, model_dir="/nonexistent", device="cuda:0")


def test_infer_with_built_in_conditioning_and_s2mel():
    """Rows N2 + N1 inside `infer`: the conditioning encoders ride in the GPT state dict, s2mel in `s2mel_state_dict`; the glue
    then only supplies prompt features and the tokenizer.  The per-request conditioning equals the CPU glue's."""
    import voice_tts_amd.conditioning as CD
    import voice_tts_amd.s2mel as S2
    import voice_tts_amd.weights as WR
    from indextts.infer_v2 import IndexTTS2

    dev = torch.device("cuda:0")
    D = 128
    gcfg = WR.tiny_gpt_cfg(model_dim=D, layers=2, heads=2)
    bcfg = WR.tiny_bigvgan_cfg(64)
    ccfg = CD.tiny_cond_cfg(model_dim=D)
    scfg = S2.tiny_s2mel_cfg(gpt_dim=D)
    Wg = dict(WR.make_gpt_weights(gcfg, seed=7))
    Wc = CD.make_cond_weights(ccfg, seed=9)
    Wg.update(Wc)
    Ws = S2.make_s2mel_weights(scfg, seed=10)

    class PromptGlue(FakeGlue):
        def speaker(self, spk_audio_prompt):
            self.calls["speaker"] += 1
            g = torch.Generator().manual_seed(21)
            return dict(spk_cond_emb=torch.randn(1, 19, ccfg["input_size"], generator=g).to(self.dev), style=torch.randn(1, scfg["style_dim"], generator=g).to(self.dev),
                        prompt_condition=torch.randn(1, 6, scfg["content_dim"], generator=g).to(self.dev), ref_mel=(torch.randn(1, 80, 6, generator=g) * 2 - 4).to(self.dev))

        def emotion(self, emo_audio_prompt):
            return torch.randn(1, 15, ccfg["input_size"], generator=torch.Generator().manual_seed(22)).to(self.dev)

        def merge_emovec(self, *a):
            raise AssertionError("the built-in conditioning must be used")

        get_conditioning = s2mel = merge_emovec

    glue = PromptGlue(D, dev)
    m = IndexTTS2(cfg_path=None, model_dir="/nonexistent", device="cuda:0", glue=glue, gpt_state_dict=Wg, bigvgan_state_dict=WR.make_bigvgan_weights(bcfg, seed=8),
                  s2mel_state_dict=Ws, gpt_cfg=gcfg, bigvgan_cfg=bcfg, cond_cfg=ccfg, s2mel_cfg=scfg, max_seq=192, max_frames=128)
    assert m.cond is not None and m.s2mel is not None
    sr, pcm = m.infer("spk.wav", "abcdefghijklmnop" * 2, None, emo_audio_prompt="emo.wav", emo_alpha=0.7, max_text_tokens_per_segment=20,
                      num_beams=1, top_k=1, max_mel_tokens=14)
    assert sr == 22050 and pcm.dtype == np.int16 and pcm.shape[0] == 2 * int(14 * 1.72) * 256 + int(22050 * 0.2)
    # the hoisted conditioning on the GPU == the CPU glue on the same prompts
    cpu = CD.Conditioning(Wc, ccfg, "cpu")
    spk, emo = glue.speaker("x")["spk_cond_emb"].cpu(), glue.emotion("y").cpu()
    ls, le = torch.tensor([spk.shape[-1]]), torch.tensor([emo.shape[-1]])
    want = cpu.get_conditioning(spk.transpose(1, 2), ls)[0] + cpu.merge_emovec(spk, emo, ls, le, alpha=0.7)
    got = m.cond.get_conditioning(spk.to(dev).transpose(1, 2), ls.to(dev))[0] + m.cond.merge_emovec(spk.to(dev), emo.to(dev), ls.to(dev), le.to(dev), alpha=0.7)
    assert (got.cpu() - want).abs().max().item() <= 2e-4 * max(1.0, want.abs().max().item())


def test_infer_tokenises_raw_text_with_the_built_in_front_end():
    """Row N4 inside `infer`: with a `TextTokenizer` the text goes through normalise -> CJK spacing -> sentencepiece ->
    `split_segments` in this package (front.py / infer_v2.py:582-617); same audio as a glue handing over the same ids."""
    import voice_tts_amd.weights as WR
    from indextts.infer_v2 import IndexTTS2
    from voice_tts_amd.front import TextNormalizer, TextTokenizer

    class Identity:
        def normalize(self, text):
            return text

    tok = TextTokenizer(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_bpe.model"),
                        TextNormalizer(zh_normalizer=Identity(), en_normalizer=Identity()))
    text = "今天是个好日子 it's a good day。最zhong4要的是：不要chong2蹈覆辙！where's the money?"
    toks = tok.tokenize(text)
    want_segments = [tok.convert_tokens_to_ids(s) for s in tok.split_segments(toks, 20)]
    assert len(want_segments) >= 3

    class IdsGlue(FakeGlue):
        def tokenize(self, text, max_text_tokens_per_segment, quick_streaming_tokens=0):
            return want_segments

    class NoTokGlue(FakeGlue):
        def tokenize(self, *a, **k):
            raise AssertionError("the built-in tokenizer must be used")

    dev = torch.device("cuda:0")
    gcfg = WR.tiny_gpt_cfg(model_dim=128, layers=2, heads=2, number_text_tokens=tok.vocab_size + 1)
    bcfg = WR.tiny_bigvgan_cfg(64)
    Wg, Wb = WR.make_gpt_weights(gcfg, seed=7), WR.make_bigvgan_weights(bcfg, seed=8)
    kw = dict(cfg_path=None, model_dir="/nonexistent", device="cuda:0", gpt_state_dict=Wg, bigvgan_state_dict=Wb, gpt_cfg=gcfg, bigvgan_cfg=bcfg,
              max_seq=192, max_frames=128)
    gen = dict(max_text_tokens_per_segment=20, num_beams=1, top_k=1, max_mel_tokens=12)
    sr_a, pcm_a = IndexTTS2(glue=NoTokGlue(128, dev), tokenizer=tok, **kw).infer("spk.wav", text, None, **gen)
    sr_b, pcm_b = IndexTTS2(glue=IdsGlue(128, dev), **kw).infer("spk.wav", text, None, **gen)
    assert sr_a == sr_b == 22050 and pcm_a.shape == pcm_b.shape and np.array_equal(pcm_a, pcm_b)


@pytest.fixture(scope="module")
def tts_from_dir(tmp_path_factory):
    """`IndexTTS2(cfg_path, model_dir)` built by its own file loaders from a synthetic model_dir -- NO glue object."""
    import synthetic_model_dir as SM
    from indextts.infer_v2 import IndexTTS2
    from voice_tts_amd.front import TextNormalizer, TextTokenizer

    root = str(tmp_path_factory.mktemp("model_dir"))
    cfg_path, cfg = SM.write_model_dir(root)

    class Same:  # WeTextProcessing is absent from the image: the verbaliser slot takes any object with .normalize (front.py)
        def normalize(self, s):
            return s

    tok = TextTokenizer(root + "/bpe.model", TextNormalizer(Same(), Same()))
    m = IndexTTS2(cfg_path=cfg_path, model_dir=root, use_fp16=False, device="cuda:0", tokenizer=tok, max_seq=256, max_frames=256)
    return m, SM


def test_infer_from_audio_bytes_and_raw_text_without_glue(tts_from_dir):
    """SURVEY 8(b) "Python API": a WAV byte string + raw text in, `(22050, int16 [N, 1])` out, every stage from model_dir
    files: audio decode -> resample -> w2v-bert features -> codec quantize -> reference mel -> fbank + CAM++ -> prompt condition
    -> conditioners -> GPT (HIP) -> latent (HIP) -> s2mel -> BigVGAN (HIP) -> PCM."""
    m, SM = tts_from_dir
    assert m.glue is None and m.ready() and not m.missing_glue
    wav = SM.synthetic_wav_bytes(1.5, 24000)
    calls = []
    real = m.prompt.speaker
    m.prompt.speaker = lambda a: (calls.append(1), real(a))[1]
    sr, pcm = m.infer(wav, "Hello world, this is a test. 你好世界！", None, num_beams=1, top_k=1, max_mel_tokens=24)
    assert sr == 22050 and pcm.dtype == np.int16 and pcm.ndim == 2 and pcm.shape[1] == 1
    assert pcm.shape[0] == int(24 * 1.72) * 256 and np.abs(pcm).max() > 0
    assert set(m.cache_spk) == {"spk_cond_emb", "style", "prompt_condition", "ref_mel"} and m.cache_spk["style"].shape == (1, 192)
    # a fresh but equal bytes object (what the server hands over per request) hits the speaker cache (infer_v2.py:508)
    sr2, pcm2 = m.infer(bytes(bytearray(wav)), "Hello world, this is a test. 你好世界！", None, num_beams=1, top_k=1, max_mel_tokens=24)
    assert len(calls) == 1 and pcm2.shape == pcm.shape
    # greedy + injected-noise-free s2mel draws torch.randn on the device: same seed state is not guaranteed, so only the
    # token-determined length is compared; a different prompt misses the cache
    m.infer(SM.synthetic_wav_bytes(1.0, 16000, seed=1), "Short.", None, num_beams=1, top_k=1, max_mel_tokens=8)
    assert len(calls) == 2
    # the other prompt forms reach the same stages: (ndarray, sr) tuple and a path on disk
    x = np.frombuffer(wav[44:], "<i2").astype(np.float32) / 32768.0
    assert m.infer((x, 24000), "Tuple prompt.", None, num_beams=1, top_k=1, max_mel_tokens=8)[1].shape[0] == int(8 * 1.72) * 256
    # emotion reference audio + alpha, then an emotion vector through the built-in matrix mix (served default: 3-beam sample)
    out = m.infer(wav, "Emotion.", None, emo_audio_prompt=SM.synthetic_wav_bytes(1.2, 16000, seed=2), emo_alpha=0.7, max_mel_tokens=10, seed=3)
    assert out[0] == 22050 and out[1].shape[0] % 256 == 0 and out[1].shape[0] > 0
    out = m.infer(wav, "Vector.", None, emo_vector=[0.3, 0, 0, 0, 0, 0, 0.2, 0.1], max_mel_tokens=10, seed=3)
    assert out[1].shape[0] % 256 == 0 and out[1].shape[0] > 0


def test_model_without_prompt_weights_reports_not_ready():
    import voice_tts_amd.weights as WR
    from indextts.infer_v2 import IndexTTS2

    gcfg, bcfg = WR.tiny_gpt_cfg(model_dim=128, layers=2, heads=2), WR.tiny_bigvgan_cfg(64)
    m = IndexTTS2(cfg_path=None, model_dir="/nonexistent", device="cuda:0", gpt_state_dict=WR.make_gpt_weights(gcfg, seed=7),
                  bigvgan_state_dict=WR.make_bigvgan_weights(bcfg, seed=8), gpt_cfg=gcfg, bigvgan_cfg=bcfg, max_seq=96, max_frames=32)
    assert not m.ready() and "campplus_cn_common.bin" in m.missing_glue
    with pytest.raises(NotImplementedError, match="campplus_cn_common.bin"):
        m.infer(b"RIFF....", "hi", None)


def test_http_tts_over_the_real_pipeline(tts_from_dir):
    """`POST /tts` end to end (server.py:320-440 surface): hex WAV in -> the full pipeline on the GPU -> hex WAV out, through the
    mirrored FastAPI app around the same `IndexTTS2` the Python-API test uses; emotion label and emotion audio variants."""
    import io

    from fastapi.testclient import TestClient

    from voice_tts_amd.server import create_app

    m, SM = tts_from_dir
    spk_hex = SM.synthetic_wav_bytes(1.2, 22050, seed=5).hex()
    with TestClient(create_app(lambda: m)) as c:
        assert c.get("/health").json() == {"status": "healthy", "model_loaded": True, "deepspeed_enabled": False}
        r = c.post("/tts", json={"text": "Hello there.", "spk_audio": spk_hex})
        assert r.status_code == 200, r.text
        body = r.json()
        assert set(body) == {"audio_hex", "audio_length", "inference_time", "rtf", "text"} and body["text"] == "Hello there."
        with wave.open(io.BytesIO(bytes.fromhex(body["audio_hex"]))) as w:
            assert (w.getnchannels(), w.getsampwidth(), w.getframerate()) == (1, 2, 22050)
            assert w.getnframes() > 0 and abs(w.getnframes() / 22050 - body["audio_length"]) < 1e-6
        assert body["rtf"] == pytest.approx(body["inference_time"] / body["audio_length"])
        r = c.post("/tts", json={"text": "Happy.", "spk_audio": spk_hex, "emotion": "高兴", "emo_alpha": 0.6})
        assert r.status_code == 200, r.text
        r = c.post("/tts", json={"text": "Emo audio.", "spk_audio": spk_hex, "emo_audio": SM.synthetic_wav_bytes(1.0, 16000, seed=6).hex(), "emo_alpha": 0.5})
        assert r.status_code == 200, r.text
        assert c.post("/tts", json={"text": "x", "spk_audio": "zz" * 80}).status_code == 400  # not hex, not a URL


def test_infer_many_shares_the_decode_slots(tts_from_dir):
    """`IndexTTS2.infer_many` (row N3 at the API level): three requests with different prompts, texts and emotion settings through ONE
    scheduler run.  With `top_k=1` the decode is deterministic: every request's audio has the length its own `infer(...,
    num_beams=1, top_k=1)` produces (same codes: a sequence's tokens do not depend on the company it decodes in), int16 mono."""
    m, SM = tts_from_dir
    wav_a, wav_b = SM.synthetic_wav_bytes(1.5, 24000), SM.synthetic_wav_bytes(1.0, 16000, seed=1)
    reqs = [dict(spk_audio_prompt=wav_a, text="Hello world, this is a test. 你好世界！"),
            dict(spk_audio_prompt=wav_b, text="Short."),
            dict(spk_audio_prompt=wav_a, text="Vector.", emo_vector=[0.3, 0, 0, 0, 0, 0, 0.2, 0.1])]
    outs = m.infer_many(reqs, decode_slots=4, num_beams=1, top_k=1, max_mel_tokens=20)
    assert len(outs) == 3 and m.last_timing["audio_length"] > 0
    for rq, (sr, pcm) in zip(reqs, outs):
        assert sr == 22050 and pcm.dtype == np.int16 and pcm.ndim == 2 and pcm.shape[1] == 1
        kw = {k: rq[k] for k in ("emo_vector",) if k in rq}
        ref_sr, ref = m.infer(rq["spk_audio_prompt"], rq["text"], None, num_beams=1, top_k=1, max_mel_tokens=20, **kw)
        assert ref.shape == pcm.shape, (rq["text"], ref.shape, pcm.shape)
    assert m.infer_many([dict(spk_audio_prompt=wav_a, text="")])[0] is None


def test_http_batched_mode_over_the_real_pipeline(tts_from_dir, monkeypatch):
    """IXTTS_BATCH_SLOTS: three concurrent POST /tts served by one `infer_many` batch on the GPU."""
    import io
    from concurrent.futures import ThreadPoolExecutor

    from fastapi.testclient import TestClient

    from voice_tts_amd.server import create_app

    m, SM = tts_from_dir
    monkeypatch.setenv("IXTTS_BATCH_SLOTS", "4")
    monkeypatch.setenv("IXTTS_BATCH_WINDOW_MS", "300")
    spk_hex = SM.synthetic_wav_bytes(1.2, 22050, seed=5).hex()
    app = create_app(lambda: m)
    with TestClient(app) as c:
        with ThreadPoolExecutor(3) as ex:
            rs = list(ex.map(lambda t: c.post("/tts", json={"text": t, "spk_audio": spk_hex}), ["One.", "Two two.", "Three three three."]))
        assert [r.status_code for r in rs] == [200, 200, 200], [r.text[:200] for r in rs]
        assert max(app.state.tts["batcher"].batches) >= 2
        for r in rs:
            with wave.open(io.BytesIO(bytes.fromhex(r.json()["audio_hex"]))) as w:
                assert w.getframerate() == 22050 and w.getnframes() > 0



@pytest.fixture(scope="module")
def tts_bf16_from_dir(tmp_path_factory):
    """The same synthetic model_dir in the throughput mode (use_fp16 -> bf16 GPT): beam groups run on the wide engine."""
    import synthetic_model_dir as SM
    from indextts.infer_v2 import IndexTTS2
    from voice_tts_amd.front import TextNormalizer, TextTokenizer

    root = str(tmp_path_factory.mktemp("model_dir_bf16"))
    cfg_path, cfg = SM.write_model_dir(root)

    class Same:
        def normalize(self, s):
            return s

    tok = TextTokenizer(root + "/bpe.model", TextNormalizer(Same(), Same()))
    m = IndexTTS2(cfg_path=cfg_path, model_dir=root, use_fp16=True, device="cuda:0", tokenizer=tok, max_seq=256, max_frames=256)
    return m, SM


def test_served_default_decodes_the_segments_beam_groups_together(tts_bf16_from_dir, monkeypatch):
    """`infer()` with its defaults (num_beams=3 beam-sample, infer_v2.py:598-606) on a text of several segments: every segment
    is a beam group and the groups step together on the wide engine (one scheduler run, 3 x segments slots); with
    IXTTS_BEAM_GROUPS=1 the segments go one after another through the register engine, as the reference runs them."""
    from voice_tts_amd import scheduler as SCH

    m, SM = tts_bf16_from_dir
    wav = SM.synthetic_wav_bytes(1.5, 24000)
    text = "Hello world, this is a test. 你好世界！ One more sentence follows here. And a last one."
    runs = []
    real_run = SCH.BeamGroupScheduler.run

    def spy(self, segments, on_done, **kw):
        st = real_run(self, segments, on_done, **kw)
        runs.append((self.engine.max_batch, self.max_groups, len(segments), dict(st)))
        return st

    monkeypatch.setattr(SCH.BeamGroupScheduler, "run", spy)
    sr, pcm = m.infer(wav, text, None, max_text_tokens_per_segment=20, max_mel_tokens=24, seed=4)
    assert sr == 22050 and pcm.dtype == np.int16 and pcm.shape[1] == 1 and pcm.shape[0] > 0
    assert len(runs) == 1 and runs[0][0] == 15 and runs[0][2] >= 3, runs  # one run, wide engine, all segments in it
    assert runs[0][3]["busy_group_steps"] > runs[0][3]["decode_calls"] * 8, runs  # several groups per step
    n_seg = runs[0][2]
    # same request, groups off: the register engine, segment after segment -- same amount of audio structure (segments + silences)
    monkeypatch.setenv("IXTTS_BEAM_GROUPS", "1")
    sr2, pcm2 = m.infer(wav, text, None, max_text_tokens_per_segment=20, max_mel_tokens=24, seed=4)
    assert len(runs) == 1 and sr2 == 22050 and pcm2.shape[0] > 0
    # every segment yields between 1 and 24 codes -> frames; 200 ms of silence between segments
    sil = (n_seg - 1) * int(22050 * 0.2)
    for p in (pcm, pcm2):
        assert sil + n_seg * 256 <= p.shape[0] <= sil + n_seg * int(24 * 1.72) * 256


def test_infer_many_keeps_the_beams_and_fails_requests_alone(tts_bf16_from_dir):
    """`infer_many` with the served defaults: segments of all requests as beam groups on the wide engine; a request whose prompt
    audio cannot be decoded gets its exception back in its slot, the others their audio."""
    m, SM = tts_bf16_from_dir
    wav_a, wav_b = SM.synthetic_wav_bytes(1.5, 24000), SM.synthetic_wav_bytes(1.0, 16000, seed=1)
    reqs = [dict(spk_audio_prompt=wav_a, text="Hello world, this is a test."),
            dict(spk_audio_prompt=b"this is not audio at all", text="Broken prompt."),
            dict(spk_audio_prompt=wav_b, text="Short."),
            dict(spk_audio_prompt=wav_a, text="")]
    outs = m.infer_many(reqs, decode_slots=9, max_mel_tokens=16, seed=2)
    assert len(outs) == 4
    assert isinstance(outs[1], Exception) and outs[3] is None
    for o in (outs[0], outs[2]):
        assert isinstance(o, tuple) and o[0] == 22050 and o[1].dtype == np.int16 and o[1].shape[0] > 0
    assert 9 in m._engines  # three groups of three beams
    # a code outside the semantic codebook is refused on the host, for that request only
    real = m.s2mel.W["quantizer.codebook.weight"]
    m.s2mel.W["quantizer.codebook.weight"] = real[:4]
    try:
        outs = m.infer_many(reqs[:1], decode_slots=9, max_mel_tokens=8, seed=2)
        assert isinstance(outs[0], ValueError) and "codebook" in str(outs[0])
    finally:
        m.s2mel.W["quantizer.codebook.weight"] = real
