"""The reference's HTTP surface (`server.py`) over this repository's `IndexTTS2` -- SURVEY.md 8(b) "HTTP API".

Same endpoints, models, priorities, error codes and CLI as the reference:
  POST /tts   TTSRequest{text, spk_audio (URL | hex > 100 chars), emo_audio?, emotion? (str | {str: float in [0,1]}), emo_alpha in [0,1] = 1.0}
              -> TTSResponse{audio_hex, audio_length, inference_time, rtf, text}                       (server.py:183-235,320-440)
              emo_audio beats emotion (:352-370); emo_alpha is forced to 1.0 unless emo_audio is given (:391)
  GET /       {"status","model_loaded","service","version"}                                            (:238-246)
  GET /health {"status","model_loaded","deepspeed_enabled"}; 503 while the model is not loaded         (:249-259)
  GET /debug/worker-info  worker id, pid, visible devices, GPU and model info                          (:262-317)
  errors: 400 unusable audio string, 408 download timeout, upstream HTTP status passed through, 500 inference failure,
          503 model not loaded                                                                         (:134-148,172-180,337-339,430-440)
  CLI: --host --port(8020) --workers --reload --log-level; workers > 1 through gunicorn with one GPU per worker (:447-551)

One inference at a time per worker process, as the reference's `inference_lock` (:25,384); `deepspeed_enabled` is always
False (the HIP decode engine stands in DeepSpeed's seam).  The model is built by `model_factory` in the worker process,
after the GPU for that worker has been chosen -- nothing touches torch at import time (server.py:17-19).
"""
import logging
import os
import re
import tempfile
import threading
import time
import wave
from contextlib import asynccontextmanager
from typing import Dict, Optional, Union

from fastapi import FastAPI, HTTPException
from fastapi.middleware.cors import CORSMiddleware
from pydantic import BaseModel, Field, field_validator

from .emotion import create_emotion_vector

logger = logging.getLogger("indextts.server")


def is_hex_string(s):
    """Hex-encoded audio: only hex digits, even length, and long enough not to be mistaken for a word (server.py:93-99)."""
    return bool(s) and bool(re.match(r"^[0-9a-fA-F]+$", s)) and len(s) % 2 == 0 and len(s) > 100


def is_url(s):
    return s.startswith(("http://", "https://", "ftp://"))


def download_audio_from_url(url, timeout=30.0):
    import requests

    try:
        logger.info(f"Downloading audio from URL: {url}")
        response = requests.get(url, timeout=timeout)
        response.raise_for_status()
        ctype = response.headers.get("content-type", "")
        if ctype and not any(t in ctype.lower() for t in ("audio", "octet-stream", "wav", "mp3", "mpeg")):
            logger.warning(f"URL content-type may not be audio: {ctype}")
        return response.content
    except requests.Timeout:
        raise HTTPException(status_code=408, detail=f"Download timeout: {url}")
    except requests.HTTPError as e:
        raise HTTPException(status_code=e.response.status_code, detail=f"Failed to download audio from URL: HTTP {e.response.status_code}")
    except Exception as e:
        raise HTTPException(status_code=500, detail=f"Error downloading audio from URL: {str(e)}")


def get_audio_data(audio_input):
    if is_url(audio_input):
        return download_audio_from_url(audio_input)
    if is_hex_string(audio_input):
        try:
            return bytes.fromhex(audio_input)
        except ValueError as e:
            raise HTTPException(status_code=400, detail=f"Invalid hex encoded audio data: {str(e)}")
    raise HTTPException(status_code=400, detail="Invalid audio input format. Must be URL (http://, https://) or hex encoded string")


class TTSRequest(BaseModel):
    text: str = Field(..., description="text to synthesise")
    spk_audio: str = Field(..., description="speaker reference audio (URL or hex)")
    emo_audio: Optional[str] = Field(None, description="emotion reference audio (URL or hex); takes priority over `emotion`")
    emotion: Optional[Union[str, Dict[str, float]]] = Field(None, description="one emotion label, or {label: strength in [0,1]}")
    emo_alpha: float = Field(default=1.0, description="emotion strength in [0,1]")

    @field_validator("emo_alpha")
    @classmethod
    def _alpha(cls, v):
        if not 0.0 <= v <= 1.0:
            raise ValueError("emo_alpha must be between 0.0 and 1.0")
        return v

    @field_validator("emotion")
    @classmethod
    def _emotion(cls, v):
        if v is None or isinstance(v, str):
            return v
        if isinstance(v, dict):
            for key, value in v.items():
                if not isinstance(key, str):
                    raise ValueError(f"Emotion dict keys must be strings, got {type(key)}")
                if not isinstance(value, (int, float)):
                    raise ValueError(f"Emotion dict values must be numbers, got {type(value)}")
                if not 0.0 <= value <= 1.0:
                    raise ValueError(f"Emotion values must be between 0.0 and 1.0, got {value}")
            return v
        raise ValueError("emotion must be a string or dict")


class TTSResponse(BaseModel):
    audio_hex: str
    audio_length: float
    inference_time: float
    rtf: float
    text: str


def broadcast_context(env=None):
    """(rank, world, port) of the load-time weight broadcast, or None.  Opt-in: IXTTS_BROADCAST_LOAD=1 with IXTTS_WORKERS=N (the
    gunicorn worker count; `main()` exports it) -- worker k (WORKER_ID = its 1-based age, set by `post_fork`) is rank k-1; only
    rank 0 reads the checkpoints, the others receive them over RCCL (xGMI) from it."""
    env = os.environ if env is None else env
    if env.get("IXTTS_BROADCAST_LOAD", "0") != "1":
        return None
    try:
        world, wid = int(env.get("IXTTS_WORKERS", "1")), int(env.get("WORKER_ID", "1"))
    except ValueError:
        return None
    if world < 2 or not 1 <= wid <= world:
        return None  # (a worker gunicorn re-forked after a crash has an age beyond the first N: it loads from the files)
    return wid - 1, world, int(env.get("IXTTS_BROADCAST_PORT", "29617"))


def cap_cpu_threads(env=None):
    """torch sizes its CPU pool by the HOST's cores; N workers on one node would each start that many threads (128 on a 16-core
    share measured: 50-100 ms hiccups on the 2 ms host-side pieces of a request -- WAV decode, resampling, the feature extractor).
    Each worker takes its share of what this process may use (cgroup quota / affinity), at most 16; IXTTS_CPU_THREADS overrides."""
    import torch

    env = os.environ if env is None else env
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    try:
        workers = max(1, int(env.get("IXTTS_WORKERS", "1")))
    except ValueError:
        workers = 1
    want = int(env["IXTTS_CPU_THREADS"]) if env.get("IXTTS_CPU_THREADS", "").isdigit() else min(16, max(1, n // workers))
    torch.set_num_threads(max(1, want))
    return torch.get_num_threads()


def default_model_factory(cfg_path="models/IndexTTS/config.yaml", model_dir="models/IndexTTS", **kw):
    """What the reference's lifespan does (server.py:66-72), on the HIP path; with IXTTS_BROADCAST_LOAD=1 the workers of the node
    form a one-shot RCCL group for the weight broadcast and dissolve it again (no steady-state collective exists)."""
    from indextts.infer_v2 import IndexTTS2

    cap_cpu_threads()
    bc = broadcast_context()
    if bc is None:
        return IndexTTS2(cfg_path=cfg_path, model_dir=model_dir, use_fp16=True, use_cuda_kernel=True, use_deepspeed=False, **kw)
    import torch
    import torch.distributed as dist

    rank, world, port = bc
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = os.environ.get("IXTTS_BROADCAST_BACKEND", "nccl")  # nccl IS RCCL on ROCm (one GPU per worker: local device 0)
    logger.info(f"Worker {rank + 1}/{world}: joining the weight broadcast group ({backend}, 127.0.0.1:{port})")
    pg_kw = dict(device_id=torch.device("cuda:0")) if backend == "nccl" else {}
    dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, **pg_kw)
    try:
        return IndexTTS2(cfg_path=cfg_path, model_dir=model_dir, use_fp16=True, use_cuda_kernel=True, use_deepspeed=False,
                         weight_broadcast=(rank, world), **kw)
    finally:
        dist.destroy_process_group()


class RequestBatcher:
    """Opt-in replacement of the per-worker inference lock (server.py:25,384) by the decode scheduler: handlers
    queue their request; one thread takes what has arrived within `window_s` (at most `slots` requests) and serves the batch
    through `model.infer_many` -- every request's segments share the decode slots, each segment a 3-beam group as in the
    lock-step path (see `IndexTTS2.infer_many`).  A request that fails (bad prompt audio ...) fails alone."""

    def __init__(self, model, slots, window_s=0.02):
        import queue
        from concurrent.futures import Future

        self.model, self.slots, self.window_s, self._Future = model, int(slots), float(window_s), Future
        self.q = queue.Queue()
        self.batches = []  # sizes of the batches served (observability / tests)
        self._t = threading.Thread(target=self._loop, name="ixtts-batcher", daemon=True)
        self._t.start()

    def submit(self, request):
        fut = self._Future()
        self.q.put((request, fut))
        return fut

    def _loop(self):
        import queue

        while True:
            batch = [self.q.get()]
            deadline = time.time() + self.window_s
            while len(batch) < self.slots:
                try:
                    batch.append(self.q.get(timeout=max(0.0, deadline - time.time())))
                except queue.Empty:
                    break
            self.batches.append(len(batch))
            try:
                results = self.model.infer_many([r for r, _ in batch], decode_slots=self.slots)
                for (_, fut), res in zip(batch, results):
                    if isinstance(res, BaseException):  # that request's own failure; the others of the batch are served
                        fut.set_exception(res)
                    else:
                        fut.set_result(res)
            except Exception as e:  # a failure of the shared decode itself fails the batch it happened in
                for _, fut in batch:
                    if not fut.done():
                        fut.set_exception(e)


def create_app(model_factory=default_model_factory):
    state = {"model": None, "batcher": None}
    inference_lock = threading.Lock()

    @asynccontextmanager
    async def lifespan(app):
        worker_id = os.environ.get("WORKER_ID", "unknown")
        visible = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES", "default"))
        logger.info(f"Worker {worker_id} (PID: {os.getpid()}) starting, GPU: {visible}")
        try:
            model = model_factory()
            # one synthetic request before /health answers 200 (IXTTS_WARMUP=0 turns it off): a fresh process pays its one-off
            # costs (library code objects, decode graphs) here, not in the first caller's request
            if os.environ.get("IXTTS_WARMUP", "1") != "0" and hasattr(model, "warm_up"):
                model.warm_up()
            state["model"] = model
            logger.info(f"Model loaded successfully on GPU: {visible}")
            slots = int(os.environ.get("IXTTS_BATCH_SLOTS", "0") or 0)
            if slots > 0 and hasattr(state["model"], "infer_many"):
                state["batcher"] = RequestBatcher(state["model"], slots, float(os.environ.get("IXTTS_BATCH_WINDOW_MS", "20")) / 1e3)
                logger.info(f"Request batching on: up to {slots} decode slots per step (IXTTS_BATCH_SLOTS)")
        except Exception as e:
            logger.error(f"Failed to load model: {e}")
            raise
        yield
        logger.info("Worker process shutting down...")

    app = FastAPI(title="IndexTTS API Server - Stateless", lifespan=lifespan)
    app.add_middleware(CORSMiddleware, allow_origins=["*"], allow_credentials=True, allow_methods=["*"], allow_headers=["*"])
    app.state.tts = state

    @app.get("/")
    def root():
        return {"status": "running", "model_loaded": state["model"] is not None, "service": "IndexTTS API Server - Stateless", "version": "2.0"}

    @app.get("/health")
    def health_check():
        if state["model"] is None:
            raise HTTPException(status_code=503, detail="Model not loaded")
        # a model whose prompt-side stages have no weights loads but cannot synthesise: that is not "healthy"
        ready = getattr(state["model"], "ready", None)
        if callable(ready) and not ready():
            raise HTTPException(status_code=503, detail="Model loaded without its prompt-side weights: "
                                + ", ".join(getattr(state["model"], "missing_glue", []) or ["text tokenizer"]))
        return {"status": "healthy", "model_loaded": True, "deepspeed_enabled": False}

    @app.get("/debug/worker-info")
    def worker_info():
        import torch

        gpu = {"cuda_available": torch.cuda.is_available(), "device_count": 0, "current_device": None, "device_name": None, "device_properties": None}
        if torch.cuda.is_available():
            gpu["device_count"] = torch.cuda.device_count()
            try:
                gpu["current_device"] = torch.cuda.current_device()
                gpu["device_name"] = torch.cuda.get_device_name(0)
                p = torch.cuda.get_device_properties(0)
                gpu["device_properties"] = {"name": p.name, "total_memory": f"{p.total_memory / 1024**3:.2f} GB", "major": p.major, "minor": p.minor}
            except Exception as e:
                gpu["error"] = str(e)
        m = state["model"]
        return {"worker_id": os.environ.get("WORKER_ID", "unknown"), "pid": os.getpid(),
                "cuda_visible_devices": os.environ.get("CUDA_VISIBLE_DEVICES", "not set"), "gpu_info": gpu,
                "model_info": {"loaded": m is not None, "device": str(m.device) if m else "not loaded", "use_fp16": m.use_fp16 if m else None,
                               "use_deepspeed": False}}

    @app.post("/tts", response_model=TTSResponse)
    def text_to_speech(request: TTSRequest):
        model = state["model"]
        if model is None:
            raise HTTPException(status_code=503, detail="Model not loaded")
        output_path = None
        try:
            logger.info(f"Processing TTS request: text='{request.text[:50]}...'")
            spk_audio_data = get_audio_data(request.spk_audio)
            emo_audio_data = emo_vector = None
            if request.emo_audio:
                emo_audio_data = get_audio_data(request.emo_audio)
            elif request.emotion:
                if isinstance(request.emotion, str):
                    emo_vector = create_emotion_vector(request.emotion, request.emo_alpha)
                else:
                    emo_vector = create_emotion_vector(request.emotion)
            start = time.time()
            if state["batcher"] is not None:  # opt-in: queued requests decode together (IXTTS_BATCH_SLOTS)
                import io

                res = state["batcher"].submit(dict(spk_audio_prompt=spk_audio_data, text=request.text, emo_audio_prompt=emo_audio_data if emo_audio_data else None,
                                                   emo_alpha=request.emo_alpha if emo_audio_data else 1.0, emo_vector=emo_vector)).result()
                if res is None:
                    raise RuntimeError("the text produced no speech segment")
                sr, pcm = res
                buf = io.BytesIO()
                with wave.open(buf, "wb") as w:
                    w.setnchannels(pcm.shape[1])
                    w.setsampwidth(2)
                    w.setframerate(sr)
                    w.writeframes(pcm.astype("<i2").tobytes())
                inference_time = time.time() - start
                audio_length = pcm.shape[0] / float(sr)
                rtf = inference_time / audio_length if audio_length > 0 else 0.0
                logger.info(f"TTS completed (batched): audio_length={audio_length:.2f}s, inference_time={inference_time:.2f}s, rtf={rtf:.4f}")
                return TTSResponse(audio_hex=buf.getvalue().hex(), audio_length=audio_length, inference_time=inference_time, rtf=rtf, text=request.text)
            if getattr(model, "returns_pcm_without_path", False):
                # The reference writes the waveform to a temporary file, reopens it for its length and reads it back for the hex
                # string (server.py:375-408).  `infer(output_path=None)` hands back (22050, int16 [N, 1]) -- the same samples the
                # file would hold -- so the RIFF container is built in memory: same bytes on the wire, no file system round trip.
                import io

                with inference_lock:  # one inference at a time per worker (server.py:25,384)
                    res = model.infer(spk_audio_prompt=spk_audio_data, text=request.text, output_path=None,
                                      emo_audio_prompt=emo_audio_data if emo_audio_data else None,
                                      emo_alpha=request.emo_alpha if emo_audio_data else 1.0, emo_vector=emo_vector, verbose=False)
                if res is None:
                    raise RuntimeError("the text produced no speech segment")
                sr, pcm = res
                buf = io.BytesIO()
                with wave.open(buf, "wb") as w:
                    w.setnchannels(pcm.shape[1])
                    w.setsampwidth(2)
                    w.setframerate(sr)
                    w.writeframes(pcm.astype("<i2").tobytes())
                inference_time = time.time() - start
                audio_length = pcm.shape[0] / float(sr)
                rtf = inference_time / audio_length if audio_length > 0 else 0.0
                audio_hex = buf.getvalue().hex()
                logger.info(f"TTS completed: audio_length={audio_length:.2f}s, inference_time={inference_time:.2f}s, rtf={rtf:.4f}, size={len(audio_hex)//2} bytes")
                return TTSResponse(audio_hex=audio_hex, audio_length=audio_length, inference_time=inference_time, rtf=rtf, text=request.text)
            with tempfile.NamedTemporaryFile(suffix=".wav", delete=False) as tmp:
                output_path = tmp.name
            with inference_lock:  # one inference at a time per worker (server.py:25,384)
                result_path = model.infer(spk_audio_prompt=spk_audio_data, text=request.text, output_path=output_path,
                                          emo_audio_prompt=emo_audio_data if emo_audio_data else None,
                                          emo_alpha=request.emo_alpha if emo_audio_data else 1.0, emo_vector=emo_vector, verbose=False)
            inference_time = time.time() - start
            with open(result_path, "rb") as f:
                audio_hex = f.read().hex()
            with wave.open(result_path, "rb") as w:
                audio_length = w.getnframes() / float(w.getframerate())
            rtf = inference_time / audio_length if audio_length > 0 else 0.0
            os.remove(output_path)
            logger.info(f"TTS completed: audio_length={audio_length:.2f}s, inference_time={inference_time:.2f}s, rtf={rtf:.4f}, size={len(audio_hex)//2} bytes")
            return TTSResponse(audio_hex=audio_hex, audio_length=audio_length, inference_time=inference_time, rtf=rtf, text=request.text)
        except HTTPException:
            raise
        except Exception as e:
            if type(e).__name__ == "UnsupportedAudioError":  # prompt.py: the built-in decoder reads RIFF/WAVE only -- the caller's input, not a server fault
                raise HTTPException(status_code=415, detail=f"Unsupported prompt audio: {str(e)}")
            logger.error(f"TTS inference failed: {str(e)}")
            try:
                if output_path and os.path.exists(output_path):
                    os.remove(output_path)
            except Exception:
                pass
            raise HTTPException(status_code=500, detail=f"TTS inference failed: {str(e)}")

    return app


def worker_gpu(worker_age, visible):
    """GPU of the worker gunicorn forked `worker_age`-th (1-based): round-robin over the visible list (gunicorn_config.py:43-60)."""
    gpus = [g.strip() for g in visible.split(",") if g.strip()] if visible else []
    return gpus[(worker_age - 1) % len(gpus)] if gpus else None


def post_fork(server, worker):
    """gunicorn hook: pin the freshly forked worker to its GPU before anything imports torch (gunicorn_config.py:43-60).
    On ROCm the runtime honours HIP_VISIBLE_DEVICES; CUDA_VISIBLE_DEVICES is set too, as the reference does."""
    visible = os.environ.get("CUDA_VISIBLE_DEVICES") or os.environ.get("HIP_VISIBLE_DEVICES") or ""
    gpu = worker_gpu(worker.age, visible)
    os.environ["WORKER_ID"] = str(worker.age)
    if gpu is not None:
        os.environ["CUDA_VISIBLE_DEVICES"] = gpu
        os.environ["HIP_VISIBLE_DEVICES"] = gpu


def main(argv=None):
    import argparse

    ap = argparse.ArgumentParser(description="IndexTTS API Server - Stateless", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    ap.add_argument("--host", type=str, default="0.0.0.0", help="Host to bind the server to")
    ap.add_argument("--port", type=int, default=8020, help="Port to bind the server to")
    ap.add_argument("--workers", type=int, default=1, help="Number of worker processes")
    ap.add_argument("--reload", action="store_true", help="Enable auto-reload for development")
    ap.add_argument("--log-level", type=str, default="info", choices=["critical", "error", "warning", "info", "debug", "trace"], help="Log level")
    args = ap.parse_args(argv)
    app = create_app()
    os.environ.setdefault("IXTTS_WORKERS", str(args.workers))  # the forked workers inherit it: world size of the opt-in load broadcast
    if args.workers > 1:
        try:
            from gunicorn.app.base import BaseApplication
        except ImportError:
            logger.error("Gunicorn is not installed. Please install it with: pip install gunicorn")
            logger.error("Or use single worker mode: python server.py --workers 1")
            raise

        class Standalone(BaseApplication):
            def load_config(self):
                for k, v in {"bind": f"{args.host}:{args.port}", "workers": args.workers, "worker_class": "uvicorn.workers.UvicornWorker",
                             "worker_connections": 1, "threads": 1, "timeout": 300, "keepalive": 5, "loglevel": args.log_level,
                             "accesslog": "-", "errorlog": "-", "preload_app": False, "post_fork": post_fork}.items():
                    self.cfg.set(k, v)

            def load(self):
                return app

        Standalone().run()
    else:
        import uvicorn

        uvicorn.run(app, host=args.host, port=args.port, reload=args.reload, log_level=args.log_level)


if __name__ == "__main__":
    main()
