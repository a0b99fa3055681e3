"""HTTP drop-in surface with a stand-in model (no GPU): routes, schemas, headers, WAV framing."""
import struct

import numpy as np
import pytest


class _FakeTTS:
    sampling_rate = 24000

    def __call__(self, text, voice="heart"):
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
    assert client.post("/v1/audio/speech", json={"input": "x", "response_format": "mp3"}).status_code == 422


def test_elevenlabs_routes(client):
    r = client.post("/v1/text-to-speech/3?output_format=pcm_24000", json={"text": "abc"})
    assert r.status_code == 200 and r.headers["x-sample-rate"] == "24000" and len(r.content) == 3 * 1920 * 2
    r = client.post("/v1/text-to-speech/3/stream", json={"text": "abc"})
    assert r.status_code == 200 and r.headers["x-sample-rate"] == "24000"
    chunks = np.frombuffer(r.content, dtype=np.float32)
    assert chunks.shape == (3 * 1920,) and np.allclose(chunks[1920:1922], 0.1)
    assert client.post("/v1/text-to-speech/3?output_format=mp3_44100_128", json={"text": "abc"}).status_code == 501
