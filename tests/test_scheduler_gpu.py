"""Continuous batching: more concurrent requests than slots, mixed lengths, blocking and streaming;
every request's audio must equal what the façade produces for it alone (greedy => deterministic)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_scheduler_matches_single_request_results():
    from smoltts_amd import SmolTTS
    from smoltts_amd.codec.synthetic import synthetic_mimi_state
    from smoltts_amd.config import GenerationSettings
    from smoltts_amd.server.scheduler import BatchScheduler
    from smoltts_amd.synthetic import named_config, synthetic_lm_state

    cfg = named_config("tiny")
    tts = SmolTTS(state=synthetic_lm_state(cfg, seed=21), config=cfg, mimi_state=synthetic_mimi_state(seed=5))
    reqs = [("first request", "heart", 6, False), ("second, longer request text", "sky", 9, True), ("3", "nova", 4, False),
            ("the fourth one", "bella", 11, False), ("fifth", "heart", 5, True), ("sixth request waits for a slot", "liam", 7, False)]
    want = []
    for text, voice, n, stream in reqs:
        gs = GenerationSettings.greedy(max_new_tokens=n)
        want.append(np.concatenate(list(tts.stream(text, voice, generation_settings=gs))) if stream else tts(text, voice, generation_settings=gs))
    # prefill_chunk=8: every prompt enters in several chunks, with decode ticks of the speaking slots in between
    sched = BatchScheduler(tts, max_batch=3, frames_per_tick=2, generation_settings=GenerationSettings.greedy(max_new_tokens=16),
                           prefill_chunk=8)
    got = [None] * len(reqs)

    def worker(i):
        text, voice, n, stream = reqs[i]
        r = sched.submit(text, voice, stream=stream, max_new_tokens=n)
        got[i] = np.concatenate(list(sched.iter_chunks(r)) or [np.zeros(0, np.float32)])

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(reqs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    sched.close()
    for i, (g, w) in enumerate(zip(got, want)):
        assert g is not None and g.shape == w.shape, (i, None if g is None else g.shape, w.shape)
        assert float(np.sqrt(np.mean((g - w) ** 2))) <= 1e-6, i
    # a request that cannot fit is answered with an error, the scheduler keeps serving
    sched2 = BatchScheduler(tts, max_batch=2, generation_settings=GenerationSettings.greedy(max_new_tokens=4))
    with pytest.raises(ValueError):
        sched2.synthesize("x" * 600)
    assert sched2.synthesize("still alive").shape[0] % 1920 == 0
    sched2.close()
