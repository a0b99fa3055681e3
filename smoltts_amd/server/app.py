"""``smoltts-server`` drop-in: the reference's HTTP surface over the HIP engine.

Routes, schemas, headers and status behaviour follow mlx_inference/src/smoltts_mlx/server/routes/
openai.py:6-28 (``POST /v1/audio/speech`` -> audio/wav attachment ``speech.wav``) and elevenlabs.py:14-63
(``POST /v1/text-to-speech/{voice_id}`` blocking with ``output_format`` pcm_*/wav_*; ``.../stream`` ->
chunked raw float32 PCM with ``X-Sample-Rate: 24000``), ``TTSCore`` (server/tts_core.py:15-84) and the
settings file (server/settings.py:12-63).  mp3 output needs pydub + ffmpeg, which the reference pulls in and this image
lacks: those formats answer 501 instead of being half-implemented.
"""
from __future__ import annotations

import argparse
import json
from typing import Literal, Optional, Union

import numpy as np
from fastapi import APIRouter, FastAPI, HTTPException, Query, Request, Response
from fastapi.responses import StreamingResponse
from pydantic import BaseModel, Field

from .wav import pcm_to_wav_bytes


class TTSCore:
    def __init__(self, model, settings=None, scheduler=None):
        self.model = model
        self.settings = settings
        self.scheduler = scheduler  # BatchScheduler: concurrent requests share the GPU batch

    def resolve_speaker_id(self, voice: Union[str, int]) -> int:
        if isinstance(voice, int):
            return voice
        if isinstance(voice, str) and voice.isnumeric():
            return int(voice)
        return 0

    def generate_audio(self, input_text: str, voice: Union[str, int], response_format: str = "wav_24000"):
        try:
            if self.scheduler is not None:
                pcm = self.scheduler.synthesize(input_text, str(voice))
            else:
                pcm = np.asarray(self.model(input_text, str(voice))).flatten()
        except ValueError as e:  # a request the engine refuses (e.g. a text too long for max_seq_len): the client's fault, not a 500
            raise HTTPException(status_code=400, detail=str(e))
        return self.format_audio_chunk(pcm, response_format)

    def stream_audio(self, input_text: str, voice: Union[str, int]):
        chunks = (self.scheduler.iter_chunks(self.scheduler.submit(input_text, str(voice), stream=True))
                  if self.scheduler is not None else self.model.stream(input_text, str(voice)))
        try:
            for chunk in chunks:
                if chunk is not None:
                    yield np.asarray(chunk, dtype=np.float32).tobytes()
        finally:
            chunks.close()  # a client that went away mid-stream: the scheduler takes its slot back (BatchScheduler.cancel)

    def format_audio_chunk(self, pcm_data: np.ndarray, output_format: str = "pcm_24000"):
        """tts_core.py:49-84: resample when the format names another rate (FFT resampling, ``scipy.signal.resample``, as
        there), then raw PCM16 (rounded, as libsndfile writes it there), WAV (io/wav.py framing) or mp3 (not available)."""
        kind, _, rate = output_format.partition("_")
        sample_rate = int(rate.split("_")[0]) if rate else 24000
        pcm_data = np.asarray(pcm_data, dtype=np.float32).reshape(-1)
        if kind not in ("pcm", "wav", "mp3"):
            raise HTTPException(status_code=400, detail=f"Format {output_format} not yet supported")
        if kind == "mp3":
            raise HTTPException(status_code=501, detail="mp3 output is not available in this build (no encoder in the image)")
        if sample_rate != 24000:
            if sample_rate <= 0:
                raise HTTPException(status_code=400, detail=f"bad sample rate in {output_format}")
            from scipy import signal

            n = int(len(pcm_data) * sample_rate / 24000)
            pcm_data = signal.resample(pcm_data, n).astype(np.float32) if n > 0 else np.zeros(0, np.float32)
        if kind == "pcm":
            return np.rint(np.clip(pcm_data, -1.0, 1.0) * 32767).astype(np.int16).tobytes(), "audio/x-pcm"
        return pcm_to_wav_bytes(pcm_data, sample_rate), "audio/wav"


class SpeechRequest(BaseModel):
    model: str = Field(default="tts-1-hd")
    input: str
    voice: Union[str, int] = Field(default="alloy")
    response_format: Literal["wav"] = Field(default="wav")


class CreateSpeechRequest(BaseModel):
    text: str
    model_id: Optional[str] = Field(default=None)


openai_router = APIRouter(prefix="/v1", tags=["OpenAI"])
eleven_router = APIRouter(prefix="/v1", tags=["ElevenLabs"])


@openai_router.post("/audio/speech")
def openai_speech(item: SpeechRequest, http_request: Request):
    core = http_request.app.state.tts_core
    audio, media_type = core.generate_audio(item.input, item.voice, item.response_format + "_24000")
    return Response(audio, media_type=media_type, headers={"Content-Disposition": 'attachment; filename="speech.wav"'})


@eleven_router.post("/text-to-speech/{voice_id}")
def text_to_speech_blocking(voice_id: str, item: CreateSpeechRequest, http_request: Request,
                                  output_format: Optional[str] = Query(None, description="pcm_<rate> | wav_<rate>")):
    core = http_request.app.state.tts_core
    fmt = output_format or "wav_24000"
    content, media_type = core.generate_audio(item.text, voice_id, fmt)
    return Response(content=content, media_type=media_type, headers={
        "Content-Disposition": f'attachment; filename="elevenlabs_speech.{fmt.split("_")[0]}"',
        "X-Sample-Rate": fmt.split("_")[1] if "_" in fmt else "24000"})


@eleven_router.post("/text-to-speech/{voice_id}/stream")
def stream_tts(voice_id: str, item: CreateSpeechRequest, http_request: Request,
                     output_format: Literal["pcm_24000"] = "pcm_24000"):
    core = http_request.app.state.tts_core
    return StreamingResponse(core.stream_audio(item.text, voice=voice_id), media_type="audio/wav", headers={
        "Content-Disposition": 'attachment; filename="speech.pcm"', "X-Sample-Rate": "24000"})


@eleven_router.get("/stats")
def stats(http_request: Request):
    """Not in the reference: serving counters (requests by outcome, frames delivered, slots in use) of the scheduler or pool."""
    sched = http_request.app.state.tts_core.scheduler
    return sched.stats() if sched is not None and hasattr(sched, "stats") else {}


def create_app(model=None, settings: Optional[dict] = None, scheduler=None) -> FastAPI:
    """``model``: a ``smoltts_amd.SmolTTS`` (or any object with ``__call__``/``stream``); ``scheduler``: an
    optional ``BatchScheduler`` so that concurrent requests are decoded together (handlers are plain
    ``def`` and run in FastAPI's thread pool; the reference's ``async def`` handlers serialise requests)."""
    from contextlib import asynccontextmanager

    @asynccontextmanager
    async def lifespan(_app):
        yield
        if scheduler is not None and hasattr(scheduler, "close"):  # finish what is in flight, then release the GPU(s)
            try:
                scheduler.close(drain=True)
            except TypeError:  # a pool: its workers drain their own schedulers
                scheduler.close()

    app = FastAPI(lifespan=lifespan)
    app.include_router(openai_router)
    app.include_router(eleven_router)
    app.state.settings = settings
    app.state.tts_core = TTSCore(model, settings, scheduler)
    return app


def main():
    import functools

    import uvicorn

    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, help="settings JSON: {checkpoint_dir, mimi_checkpoint, generation{...}, model_type{...}}")
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--gpus", type=int, default=1, help="worker processes, one per GPU (request-level data parallelism); 1 = serve from this process")
    args = ap.parse_args()
    from .pool import GpuPool, scheduler_from_settings
    from .settings import ServerSettings

    try:
        settings = ServerSettings.get_settings(args.config)
        settings.get_checkpoint_dir()
    except ValueError as e:
        raise SystemExit(f"settings: {e}")
    settings = settings.model_dump()  # plain data: it travels to the worker processes

    if args.gpus > 1:
        # this process stays off the GPUs: it parses HTTP and relays audio; every worker owns one GPU and one scheduler
        sched = GpuPool(functools.partial(scheduler_from_settings, settings), devices=list(range(args.gpus)))
        model = None
    else:
        sched = scheduler_from_settings(settings)
        model = sched.tts
    uvicorn.run(create_app(model, settings, sched), host="0.0.0.0", port=args.port)


if __name__ == "__main__":
    main()
