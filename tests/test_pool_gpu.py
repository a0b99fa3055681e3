"""Serving front-end over GPU worker processes: two replicas (both on GPU 0 — the box has one) behind one pool and the
HTTP routes; every answer must equal what the single-request façade computes in this process."""
import functools
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_pool_of_gpu_workers_matches_the_facade_and_serves_http():
    pytest.importorskip("httpx")
    from fastapi.testclient import TestClient

    from smoltts_amd import SmolTTS
    from smoltts_amd.codec.synthetic import synthetic_mimi_state
    from smoltts_amd.config import GenerationSettings
    from smoltts_amd.server.app import create_app
    from smoltts_amd.server.pool import GpuPool, synthetic_scheduler
    from smoltts_amd.synthetic import named_config, synthetic_lm_state

    pool = GpuPool(functools.partial(synthetic_scheduler, "tiny", 21, 5, 2, 2, 16), devices=[0, 0], start_method="forkserver")
    try:
        cfg = named_config("tiny")
        tts = SmolTTS(state=synthetic_lm_state(cfg, seed=21), config=cfg, mimi_state=synthetic_mimi_state(seed=5))
        reqs = [("first request", "heart", 6, False), ("second, longer request text", "sky", 9, True), ("3", "nova", 4, False),
                ("the fourth one", "bella", 11, True), ("fifth", "heart", 5, False), ("sixth request", "liam", 7, False)]
        want = []
        for text, voice, n, stream in reqs:
            gs = GenerationSettings.greedy(max_new_tokens=n)
            want.append(np.concatenate(list(tts.stream(text, voice, generation_settings=gs))) if stream else tts(text, voice, generation_settings=gs))
        got, workers = [None] * len(reqs), [None] * len(reqs)

        def client(i):
            text, voice, n, stream = reqs[i]
            r = pool.submit(text, voice, stream=stream, max_new_tokens=n)
            workers[i] = r.worker
            got[i] = np.concatenate(list(pool.iter_chunks(r)) or [np.zeros(0, np.float32)])

        threads = [threading.Thread(target=client, args=(i,)) for i in range(len(reqs))]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
        for i, (g, w) in enumerate(zip(got, want)):
            assert g is not None and g.shape == w.shape, (i, None if g is None else g.shape, w.shape)
            assert float(np.sqrt(np.mean((g - w) ** 2))) <= 1e-6, i
        assert set(workers) == {0, 1}
        # the HTTP routes on top of the pool (no model object in the front-end process)
        client_http = TestClient(create_app(None, {}, pool))
        full = tts("over http", "0", generation_settings=GenerationSettings.greedy(max_new_tokens=16))
        resp = client_http.post("/v1/audio/speech", json={"input": "over http", "voice": "0"})
        assert resp.status_code == 200 and resp.content[:4] == b"RIFF"
        pcm16, ref16 = np.frombuffer(resp.content[44:], dtype=np.int16), (full * 32767).astype(np.int16)
        assert pcm16.shape == ref16.shape and int(np.abs(pcm16.astype(np.int32) - ref16.astype(np.int32)).max()) <= 1
        resp = client_http.post("/v1/text-to-speech/0/stream", json={"text": "over http"})
        streamed = np.frombuffer(resp.content, dtype=np.float32)
        want_stream = np.concatenate(list(tts.stream("over http", "0", generation_settings=GenerationSettings.greedy(max_new_tokens=16))))
        assert streamed.shape == want_stream.shape and float(np.sqrt(np.mean((streamed - want_stream) ** 2))) <= 1e-6
    finally:
        pool.close()
