"""HTTP drop-in surface with a stand-in model (no GPU): routes, schemas, headers, WAV framing."""
import struct

import numpy as np
import pytest


class _FakeTTS:
    sampling_rate = 24000

    def __call__(self, text, voice="heart"):
        if len(text) > 1000:
            raise ValueError("prompt + max_new_tokens exceed max_seq_len")
        return np.linspace(-0.5, 0.5, 1920 * max(1, len(text)), dtype=np.float32)

    def stream(self, text, voice="heart"):
        for i in range(3):
            yield np.full(1920, 0.1 * i, dtype=np.float32)


@pytest.fixture()
def client():
    pytest.importorskip("httpx")
    from fastapi.testclient import TestClient

    from smoltts_amd.server.app import create_app

    return TestClient(create_app(_FakeTTS()))


def test_wav_header_matches_reference_layout():
    from smoltts_amd.server.wav import pcm_to_wav_bytes

    pcm = np.array([0.0, 0.5, -0.5, 1.0], dtype=np.float32)
    b = pcm_to_wav_bytes(pcm, 24000)
    assert len(b) == 44 + 8 and b[:4] == b"RIFF" and b[8:16] == b"WAVEfmt "
    assert struct.unpack("<I", b[4:8])[0] == 8 + 36 and struct.unpack("<HHIIHH", b[20:36]) == (1, 1, 24000, 48000, 2, 16)
    assert b[36:40] == b"data" and struct.unpack("<I", b[40:44])[0] == 8
    assert np.frombuffer(b[44:], dtype=np.int16).tolist() == [0, 16383, -16383, 32767]


def test_openai_speech_route(client):
    r = client.post("/v1/audio/speech", json={"model": "tts-1-hd", "input": "hi", "voice": "heart"})
    assert r.status_code == 200 and r.headers["content-type"] == "audio/wav"
    assert r.headers["content-disposition"] == 'attachment; filename="speech.wav"'
    assert r.content[:4] == b"RIFF" and len(r.content) == 44 + 2 * 1920 * 2
    assert client.post("/v1/audio/speech", json={"voice": "heart"}).status_code == 422          # missing input
    r = client.post("/v1/audio/speech", json={"input": "x" * 2000})                               # refused by the engine
    assert r.status_code == 400 and "max_seq_len" in r.json()["detail"]
    assert client.post("/v1/audio/speech", json={"input": "x", "response_format": "mp3"}).status_code == 422


def test_elevenlabs_routes(client):
    r = client.post("/v1/text-to-speech/3?output_format=pcm_24000", json={"text": "abc"})
    assert r.status_code == 200 and r.headers["x-sample-rate"] == "24000" and len(r.content) == 3 * 1920 * 2
    r = client.post("/v1/text-to-speech/3/stream", json={"text": "abc"})
    assert r.status_code == 200 and r.headers["x-sample-rate"] == "24000"
    chunks = np.frombuffer(r.content, dtype=np.float32)
    assert chunks.shape == (3 * 1920,) and np.allclose(chunks[1920:1922], 0.1)
    assert client.post("/v1/text-to-speech/3?output_format=mp3_44100_128", json={"text": "abc"}).status_code == 501
    # other sample rates are FFT-resampled as in the reference (tts_core.py:56-59)
    r = client.post("/v1/text-to-speech/3?output_format=pcm_16000", json={"text": "abc"})
    assert r.status_code == 200 and r.headers["x-sample-rate"] == "16000" and len(r.content) == 2 * (3 * 1920 * 16000 // 24000)
    r = client.post("/v1/text-to-speech/3?output_format=wav_44100", json={"text": "abc"})
    assert r.status_code == 200 and r.content[:4] == b"RIFF" and struct.unpack("<I", r.content[24:28])[0] == 44100
    assert len(r.content) == 44 + 2 * int(3 * 1920 * 44100 / 24000)
    assert client.post("/v1/text-to-speech/3?output_format=flac_24000", json={"text": "abc"}).status_code == 400
    assert client.get("/v1/stats").json() == {}  # no scheduler behind this app: nothing to count


def test_resampling_follows_scipy_fft_resample():
    from scipy import signal

    from smoltts_amd.server.app import TTSCore

    t = np.arange(4800) / 24000.0
    pcm = (0.5 * np.sin(2 * np.pi * 440 * t)).astype(np.float32)
    data, media = TTSCore(None).format_audio_chunk(pcm, "pcm_8000")
    got = np.frombuffer(data, dtype=np.int16)
    want = np.rint(np.clip(signal.resample(pcm, 1600), -1, 1) * 32767).astype(np.int16)
    assert media == "audio/x-pcm" and got.shape == want.shape and int(np.abs(got.astype(int) - want.astype(int)).max()) <= 1
    # still a 440 Hz tone of the same amplitude
    spec = np.abs(np.fft.rfft(got.astype(np.float64)))
    assert abs(int(spec.argmax()) * 8000 / 1600 - 440) <= 5 and abs(np.abs(got).max() / 32767 - 0.5) < 0.01


def test_settings_schema_follows_the_reference(tmp_path):
    """server/settings.py:12-25: exactly one model source; generation defaults 0.5 / 0.0 / 0.10 / 1024 (:33-38)."""
    import json

    from smoltts_amd.server.settings import ServerSettings

    p = tmp_path / "config.json"
    p.write_text(json.dumps({"checkpoint_dir": "/ckpt", "model_type": {"family": "dual_ar", "codec": "mimi", "version": None},
                             "generation": {"default_temp": 0.3, "default_fast_temp": 0.0, "min_p": 0.05, "max_new_tokens": 512},
                             "mimi_checkpoint": "/mimi", "weight_format": "fp8"}))
    st = ServerSettings.get_settings(str(p))
    gs = st.generation.to_settings()
    assert str(st.get_checkpoint_dir()) == "/ckpt" and (gs.default_temp, gs.default_fast_temp, gs.min_p, gs.max_new_tokens) == (0.3, 0.0, 0.05, 512)
    assert st.weight_format == "fp8" and st.max_batch == 32
    d = ServerSettings(checkpoint_dir="/ckpt").generation
    assert (d.default_temp, d.default_fast_temp, d.min_p, d.max_new_tokens) == (0.5, 0.0, 0.10, 1024)
    with pytest.raises(ValueError, match="both"):
        ServerSettings(model_id="jkeisling/smoltts_v0", checkpoint_dir="/ckpt")
    with pytest.raises(ValueError, match="either"):
        ServerSettings()
    with pytest.raises(ValueError, match="network"):
        ServerSettings(model_id="jkeisling/smoltts_v0").get_checkpoint_dir()
    with pytest.raises(ValueError):
        ServerSettings(checkpoint_dir="/ckpt", model_type={"family": "fish", "codec": "1.4", "version": "1.4"})
    with pytest.raises(ValueError):
        ServerSettings.get_settings(None)
    assert ServerSettings(**st.model_dump()) == st  # what travels to the worker processes rebuilds the same settings
