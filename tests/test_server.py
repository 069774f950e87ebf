"""Outer boundary, HTTP side: the /tts surface of the reference's server.py over a stub model (no GPU)."""
import io
import wave

import numpy as np
import pytest
from fastapi.testclient import TestClient

from voice_tts_amd import emotion as EM
from voice_tts_amd.server import create_app, is_hex_string, worker_gpu


class StubTTS:
    device, use_fp16 = "cuda:0", True

    def __init__(self):
        self.calls = []

    def infer(self, spk_audio_prompt, text, output_path, emo_audio_prompt=None, emo_alpha=1.0, emo_vector=None, verbose=False):
        self.calls.append(dict(spk=spk_audio_prompt, text=text, emo=emo_audio_prompt, alpha=emo_alpha, vec=emo_vector))
        if text == "boom":
            raise RuntimeError("kernel exploded")
        pcm = (np.arange(22050) % 100).astype("<i2")
        with wave.open(output_path, "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(22050); f.writeframes(pcm.tobytes())
        return output_path


HEX = "00ff" * 60  # 240 hex chars: long enough to count as audio


@pytest.fixture
def client():
    stub = StubTTS()
    with TestClient(create_app(lambda: stub)) as c:
        yield c, stub


def test_status_endpoints(client):
    c, _ = client
    assert c.get("/").json() == {"status": "running", "model_loaded": True, "service": "IndexTTS API Server - Stateless", "version": "2.0"}
    assert c.get("/health").json() == {"status": "healthy", "model_loaded": True, "deepspeed_enabled": False}
    info = c.get("/debug/worker-info").json()
    assert set(info) == {"worker_id", "pid", "cuda_visible_devices", "gpu_info", "model_info"}
    assert info["model_info"] == {"loaded": True, "device": "cuda:0", "use_fp16": True, "use_deepspeed": False}


def test_health_is_503_before_the_model_is_loaded():
    app = create_app(lambda: None)
    c = TestClient(app)  # no lifespan: nothing loaded
    assert c.get("/health").status_code == 503
    assert c.get("/").json()["model_loaded"] is False
    assert c.post("/tts", json={"text": "hi", "spk_audio": HEX}).status_code == 503


def test_tts_round_trip_and_emotion_priorities(client):
    c, stub = client
    r = c.post("/tts", json={"text": "hello", "spk_audio": HEX})
    assert r.status_code == 200
    body = r.json()
    assert set(body) == {"audio_hex", "audio_length", "inference_time", "rtf", "text"} and body["text"] == "hello"
    wav = wave.open(io.BytesIO(bytes.fromhex(body["audio_hex"])))
    assert (wav.getnchannels(), wav.getsampwidth(), wav.getframerate(), wav.getnframes()) == (1, 2, 22050, 22050)
    assert body["audio_length"] == pytest.approx(1.0) and body["rtf"] == pytest.approx(body["inference_time"] / 1.0)
    assert stub.calls[-1] == dict(spk=bytes.fromhex(HEX), text="hello", emo=None, alpha=1.0, vec=None)
    # emotion label: alpha goes into the vector, the call's emo_alpha is forced to 1.0 (server.py:391)
    c.post("/tts", json={"text": "a", "spk_audio": HEX, "emotion": "高兴", "emo_alpha": 0.6})
    assert stub.calls[-1]["vec"] == [0.6, 0, 0, 0, 0, 0, 0, 0] and stub.calls[-1]["alpha"] == 1.0 and stub.calls[-1]["emo"] is None
    # emotion dict: emo_alpha ignored
    c.post("/tts", json={"text": "a", "spk_audio": HEX, "emotion": {"happy": 0.7, "生气": 0.3}, "emo_alpha": 0.2})
    assert stub.calls[-1]["vec"] == [0.7, 0.3, 0, 0, 0, 0, 0, 0]
    # emo_audio beats emotion and carries emo_alpha (server.py:352-370,391)
    c.post("/tts", json={"text": "a", "spk_audio": HEX, "emo_audio": "ab" * 80, "emotion": "sad", "emo_alpha": 0.4})
    assert stub.calls[-1]["emo"] == bytes.fromhex("ab" * 80) and stub.calls[-1]["alpha"] == 0.4 and stub.calls[-1]["vec"] is None


def test_tts_errors(client, monkeypatch):
    c, _ = client
    assert c.post("/tts", json={"text": "a", "spk_audio": "not audio"}).status_code == 400
    assert c.post("/tts", json={"text": "a", "spk_audio": "abcd"}).status_code == 400           # hex, but too short to be audio
    assert c.post("/tts", json={"text": "a", "spk_audio": HEX, "emo_alpha": 1.5}).status_code == 422
    assert c.post("/tts", json={"text": "a", "spk_audio": HEX, "emotion": {"happy": 2}}).status_code == 422
    r = c.post("/tts", json={"text": "boom", "spk_audio": HEX})
    assert r.status_code == 500 and "kernel exploded" in r.json()["detail"]

    import requests

    class Resp:
        status_code, headers, content = 404, {}, b""

        def raise_for_status(self):
            raise requests.HTTPError(response=self)

    monkeypatch.setattr(requests, "get", lambda url, timeout: Resp())
    assert c.post("/tts", json={"text": "a", "spk_audio": "https://example.invalid/x.wav"}).status_code == 404  # upstream status passed through

    def slow(url, timeout):
        raise requests.Timeout()

    monkeypatch.setattr(requests, "get", slow)
    assert c.post("/tts", json={"text": "a", "spk_audio": "http://example.invalid/x.wav"}).status_code == 408


def test_emotion_known_answers_and_helpers():
    # the reference's docstring examples (emotion.py:269-273,300-304)
    assert EM.create_emotion_vector("happy", 0.8) == [0.8, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
    assert EM.create_emotion_vector({"高兴": 0.7, "平静": 0.3}) == [0.7, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.3]
    assert EM.normalize_emotion_label(" Joyful ") == "happy" and EM.normalize_emotion_label("生气") == "angry"
    assert EM.normalize_emotion_label("???") == "calm"
    assert EM.normalize_emotion_dict({"happy": 0.7, "joyful": 0.5})["happy"] == 0.7  # one dimension named twice keeps the larger value
    assert is_hex_string("ab" * 51) and not is_hex_string("ab" * 50) and not is_hex_string("zz" * 60) and not is_hex_string("abc" * 41)
    # gunicorn worker -> GPU, round robin over the visible list (gunicorn_config.py:53-54)
    assert [worker_gpu(a, "0,1,2") for a in (1, 2, 3, 4)] == ["0", "1", "2", "0"] and worker_gpu(1, "") is None


def test_emotion_vocabulary_equals_the_reference_for_every_label():
    """tests/golden/emotion_labels.json = the reference's EMOTION_MAPPING and create_emotion_vector answers (make_golden.py
    `gen_emotion`): the label -> dimension table is wire behaviour, so all 119 keys must agree and nothing else be accepted."""
    import json
    import os

    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "emotion_labels.json"), encoding="utf-8"))
    assert EM.STANDARD_EMOTION_ORDER == g["order"]
    assert EM.EMOTION_MAPPING == g["mapping"] and len(g["mapping"]) == 119
    for c in g["label_cases"]:
        assert EM.normalize_emotion_label(c["label"]) == c["standard"], c
    for c in g["string_cases"]:
        assert EM.create_emotion_vector(c["label"], c["alpha"]) == c["vector"], c
    for c in g["dict_cases"]:
        assert EM.create_emotion_vector(c["input"]) == c["vector"], c


def test_opt_in_request_batching_replaces_the_lock(monkeypatch):
    """IXTTS_BATCH_SLOTS=N (SURVEY 8(f) N3 at the server): concurrent /tts requests are served through `model.infer_many` in
    batches instead of one at a time behind the lock; same response model, a failing batch answers 500."""
    import threading
    import time
    from concurrent.futures import ThreadPoolExecutor

    class BatchStub(StubTTS):
        def __init__(self):
            super().__init__()
            self.batches, self.gate = [], threading.Event()

        def infer_many(self, requests, decode_slots=8, **kw):
            self.batches.append([r["text"] for r in requests])
            if any(r["text"] == "boom" for r in requests):
                raise RuntimeError("kernel exploded")
            time.sleep(0.05)
            return [(22050, (np.arange(2205 * (1 + i)) % 100).astype(np.int16).reshape(-1, 1)) for i, _ in enumerate(requests)]

    monkeypatch.setenv("IXTTS_BATCH_SLOTS", "4")
    monkeypatch.setenv("IXTTS_BATCH_WINDOW_MS", "150")
    stub = BatchStub()
    with TestClient(create_app(lambda: stub)) as c:
        with ThreadPoolExecutor(3) as ex:
            rs = list(ex.map(lambda t: c.post("/tts", json={"text": t, "spk_audio": HEX, "emotion": "sad"}), ["a", "b", "c"]))
        assert [r.status_code for r in rs] == [200, 200, 200]
        assert sorted(sum(stub.batches, [])) == ["a", "b", "c"] and max(len(b) for b in stub.batches) >= 2 and not stub.calls  # infer() never ran
        for r in rs:
            body = r.json()
            with wave.open(io.BytesIO(bytes.fromhex(body["audio_hex"]))) as w:
                assert (w.getnchannels(), w.getsampwidth(), w.getframerate()) == (1, 2, 22050)
                assert abs(w.getnframes() / 22050 - body["audio_length"]) < 1e-9
        assert c.post("/tts", json={"text": "boom", "spk_audio": HEX}).status_code == 500


def test_batched_requests_fail_alone(monkeypatch):
    """One malformed request in a batch (its entry of `infer_many`'s result is the exception it raised) answers 500; the two
    good requests batched with it are served."""
    import threading
    from concurrent.futures import ThreadPoolExecutor

    class BatchStub(StubTTS):
        def __init__(self):
            super().__init__()
            self.batches = []

        def infer_many(self, requests, decode_slots=8, **kw):
            self.batches.append([r["text"] for r in requests])
            return [ValueError("speaker audio is not RIFF/WAVE") if r["text"] == "bad" else (22050, np.zeros((2205, 1), np.int16)) for r in requests]

    monkeypatch.setenv("IXTTS_BATCH_SLOTS", "4")
    monkeypatch.setenv("IXTTS_BATCH_WINDOW_MS", "300")
    stub = BatchStub()
    with TestClient(create_app(lambda: stub)) as c:
        barrier = threading.Barrier(3)

        def post(t):
            barrier.wait()
            return c.post("/tts", json={"text": t, "spk_audio": HEX})

        with ThreadPoolExecutor(3) as ex:
            rs = dict(zip(["good1", "bad", "good2"], ex.map(post, ["good1", "bad", "good2"])))
        assert any(len(b) == 3 for b in stub.batches), stub.batches  # the three really shared a batch
        assert rs["good1"].status_code == 200 and rs["good2"].status_code == 200
        assert rs["bad"].status_code == 500 and "RIFF" in rs["bad"].text


def test_worker_warms_up_before_it_reports_healthy(monkeypatch):
    """The lifespan runs `model.warm_up()` (one synthetic request) before the model is published: /health can only answer 200
    afterwards; IXTTS_WARMUP=0 skips it."""
    class Warm(StubTTS):
        def __init__(self):
            super().__init__()
            self.warmed = 0

        def warm_up(self):
            self.warmed += 1
            return 0.01

    m = Warm()
    with TestClient(create_app(lambda: m)) as c:
        assert m.warmed == 1 and c.get("/health").status_code == 200
    monkeypatch.setenv("IXTTS_WARMUP", "0")
    m2 = Warm()
    with TestClient(create_app(lambda: m2)) as c:
        assert m2.warmed == 0 and c.get("/health").status_code == 200


def test_unsupported_prompt_audio_is_a_client_error_and_pcm_models_skip_the_temp_file():
    """`prompt.UnsupportedAudioError` (the built-in decoder reads RIFF/WAVE only) -> 415 naming the container; a model that
    returns `(sr, pcm)` for `output_path=None` is served without the temporary-file round trip, same bytes on the wire."""
    from voice_tts_amd.prompt import UnsupportedAudioError

    class PcmTTS(StubTTS):
        returns_pcm_without_path = True

        def infer(self, spk_audio_prompt, text, output_path, **kw):
            assert output_path is None
            if text == "mp3":
                raise UnsupportedAudioError("prompt audio is not a RIFF/WAVE stream: this build decodes WAV only")
            return 22050, (np.arange(4410) % 100).astype(np.int16).reshape(-1, 1)

    with TestClient(create_app(lambda: PcmTTS())) as c:
        r = c.post("/tts", json={"text": "mp3", "spk_audio": HEX})
        assert r.status_code == 415 and "WAV" in r.json()["detail"]
        r = c.post("/tts", json={"text": "ok", "spk_audio": HEX})
        assert r.status_code == 200
        body = r.json()
        raw = bytes.fromhex(body["audio_hex"])
        assert raw[:4] == b"RIFF" and len(raw) == 44 + 2 * 4410 and abs(body["audio_length"] - 0.2) < 1e-9
        with wave.open(io.BytesIO(raw)) as w:
            assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 22050, 4410)


def test_worker_cpu_pool_is_the_process_share_not_the_host(monkeypatch):
    """torch sizes its CPU pool by the host's cores; a worker takes its share of what the process may use (at most 16), split over
    IXTTS_WORKERS, and IXTTS_CPU_THREADS overrides (server.cap_cpu_threads, called by the model factory)."""
    import os

    import torch

    from voice_tts_amd import server

    before = torch.get_num_threads()
    try:
        share = len(os.sched_getaffinity(0))
        assert server.cap_cpu_threads({"IXTTS_WORKERS": "1"}) == min(16, share) or share > 16
        assert server.cap_cpu_threads({"IXTTS_WORKERS": "8"}) == max(1, min(16, share // 8)) or share > 128
        assert server.cap_cpu_threads({"IXTTS_WORKERS": "bogus", "IXTTS_CPU_THREADS": "3"}) == 3
        assert 1 <= server.cap_cpu_threads({}) <= 16
    finally:
        torch.set_num_threads(before)
