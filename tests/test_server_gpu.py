"""The HTTP drop-in surface end to end on the GPU: real engine + continuous-batching scheduler behind the reference routes,
concurrent clients; every answer equals what the façade produces for that request alone."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_routes_with_scheduler_and_concurrent_clients():
    pytest.importorskip("httpx")
    from fastapi.testclient import TestClient

    from smoltts_amd import SmolTTS
    from smoltts_amd.codec.synthetic import synthetic_mimi_state
    from smoltts_amd.config import GenerationSettings
    from smoltts_amd.server.app import create_app
    from smoltts_amd.server.scheduler import BatchScheduler
    from smoltts_amd.synthetic import named_config, synthetic_lm_state

    cfg = named_config("tiny")
    tts = SmolTTS(state=synthetic_lm_state(cfg, seed=31), config=cfg, mimi_state=synthetic_mimi_state(seed=6))
    gs = GenerationSettings.greedy(max_new_tokens=10)
    sched = BatchScheduler(tts, max_batch=4, frames_per_tick=3, generation_settings=gs)
    client = TestClient(create_app(tts, scheduler=sched))
    texts = ["one", "the second request", "3", "number four is a little longer", "five", "six six six"]
    want_pcm = [tts(t, "alloy", generation_settings=gs) for t in texts]  # unknown voice name -> speaker 0, as in the reference
    want_stream = [np.concatenate(list(tts.stream(t, "2", generation_settings=gs))) for t in texts[:3]]
    got = {}

    def speech(i):
        r = client.post("/v1/audio/speech", json={"model": "tts-1-hd", "input": texts[i], "voice": "alloy"})
        got[("speech", i)] = r

    def stream(i):
        r = client.post("/v1/text-to-speech/2/stream", json={"text": texts[i]})
        got[("stream", i)] = r

    threads = [threading.Thread(target=speech, args=(i,)) for i in range(len(texts))] + [threading.Thread(target=stream, args=(i,)) for i in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    st = client.get("/v1/stats").json()
    assert st["completed"] == len(texts) + 3 and st["failed"] == 0 and st["slots"] == 4 and st["frames_delivered"] > 0
    sched.close()
    for i, w in enumerate(want_pcm):
        r = got[("speech", i)]
        assert r.status_code == 200 and r.headers["content-type"] == "audio/wav" and r.content[:4] == b"RIFF"
        pcm16 = np.frombuffer(r.content[44:], dtype=np.int16)
        ref16 = (w * 32767).astype(np.int16)
        assert pcm16.shape == ref16.shape and int(np.abs(pcm16.astype(np.int32) - ref16.astype(np.int32)).max()) <= 1
    for i, w in enumerate(want_stream):
        r = got[("stream", i)]
        assert r.status_code == 200 and r.headers["x-sample-rate"] == "24000"
        chunks = np.frombuffer(r.content, dtype=np.float32)
        assert chunks.shape == w.shape and float(np.sqrt(np.mean((chunks - w) ** 2))) <= 1e-6
