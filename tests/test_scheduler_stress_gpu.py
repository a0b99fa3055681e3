"""Scheduler under churn: many short requests with random budgets, arrival times, response kinds and cancellations through
few slots.  Every request that was not cancelled must receive exactly the single-request audio (greedy => deterministic),
cancelled ones must end, and the scheduler must come out empty."""
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_scheduler_survives_churn_and_stays_exact():
    from smoltts_amd import SmolTTS
    from smoltts_amd.codec.synthetic import synthetic_mimi_state
    from smoltts_amd.config import GenerationSettings
    from smoltts_amd.server.scheduler import BatchScheduler
    from smoltts_amd.synthetic import named_config, synthetic_lm_state

    cfg = named_config("tiny")
    tts = SmolTTS(state=synthetic_lm_state(cfg, seed=21), config=cfg, mimi_state=synthetic_mimi_state(seed=5))
    rng = np.random.default_rng(7)
    texts = ["a", "two words", "the third text is longer than the others", "4444", "five five", "six"]
    voices = ["heart", "sky", "nova"]
    NMAX = 24
    cache = {}

    def want(text, voice, stream, n):
        """The single-request façade at this budget (blocking answers drop non-semantic frames, so lengths differ)."""
        key = (text, voice, stream, n)
        if key not in cache:
            gs = GenerationSettings.greedy(max_new_tokens=n)
            cache[key] = (np.concatenate(list(tts.stream(text, voice, generation_settings=gs))) if stream
                          else tts(text, voice, generation_settings=gs))
        return cache[key]

    n_req = 120
    plan = [(texts[rng.integers(len(texts))], voices[rng.integers(len(voices))], bool(rng.integers(2)), int(rng.integers(1, NMAX + 1)),
             float(rng.uniform(0, 0.4)), rng.random() < 0.2) for _ in range(n_req)]
    sched = BatchScheduler(tts, max_batch=3, frames_per_tick=3, generation_settings=GenerationSettings.greedy(max_new_tokens=NMAX),
                           prefill_chunk=16)
    results = [None] * n_req
    errors = []

    def client(i):
        text, voice, stream, n, delay, cancel = plan[i]
        try:
            time.sleep(delay)
            r = sched.submit(text, voice, stream=stream, max_new_tokens=n)
            if cancel:
                it = sched.iter_chunks(r)
                got = []
                if stream and n > 3:
                    got.append(next(it))
                it.close()  # hang up: before the first chunk (queued or speaking) or after it
                if not got:
                    sched.cancel(r)  # a generator that never started has no finally to run
                results[i] = ("cancelled", got)
            else:
                results[i] = ("done", np.concatenate(list(sched.iter_chunks(r)) or [np.zeros(0, np.float32)]))
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=client, args=(i,)) for i in range(n_req)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=180)
    assert not errors, errors[:3]
    assert all(not t.is_alive() for t in threads), "a client is stuck"
    # everything handed back: slots free, nothing queued, nothing waiting for a codec pass
    deadline = time.time() + 20
    while (sched._active or sched._finished or sched._deliveries or not sched._pending.empty()) and time.time() < deadline:
        time.sleep(0.05)
    assert not sched._active and not sched._finished and not sched._deliveries and sorted(sched._free) == [0, 1, 2]
    sched.close()
    n_checked = 0
    for (text, voice, stream, n, _, cancel), res in zip(plan, results):
        assert res is not None
        kind, got = res
        if kind == "cancelled":
            for c in got:  # what did arrive before the hang-up is still the right audio
                w = want(text, voice, True, n)[: c.shape[0]]
                assert float(np.sqrt(np.mean((c - w) ** 2))) <= 1e-6
            continue
        w = want(text, voice, stream, n)
        assert got.shape == w.shape, (text, voice, stream, n, got.shape, w.shape)
        assert float(np.sqrt(np.mean((got - w) ** 2))) <= 1e-6
        n_checked += 1
    assert n_checked >= 80


def test_blocking_codec_pool_refills_and_stays_exact():
    """The codec slot pool of the blocking requests with a tiny geometry (3 slots, 4 frames per pass): utterances need many
    passes, finish at different times, leave holes and are replaced mid-stream; the audio must not change."""
    from smoltts_amd import SmolTTS
    from smoltts_amd.codec.synthetic import synthetic_mimi_state
    from smoltts_amd.config import GenerationSettings
    from smoltts_amd.server.scheduler import BatchScheduler
    from smoltts_amd.synthetic import named_config, synthetic_lm_state

    class SmallPool(BatchScheduler):
        CODEC_BATCH, CODEC_CHUNK, CODEC_WAIT = 3, 4, 2

    cfg = named_config("tiny")
    tts = SmolTTS(state=synthetic_lm_state(cfg, seed=21), config=cfg, mimi_state=synthetic_mimi_state(seed=5))
    budgets = [23, 2, 9, 17, 1, 30, 5, 12, 26, 3, 8, 21]
    texts = [f"utterance number {i} " + "x" * (i % 5) for i in range(len(budgets))]
    want = [tts(t, "sky", generation_settings=GenerationSettings.greedy(max_new_tokens=n)) for t, n in zip(texts, budgets)]
    sched = SmallPool(tts, max_batch=4, frames_per_tick=3, generation_settings=GenerationSettings.greedy(max_new_tokens=32))
    got = [None] * len(budgets)

    def client(i):
        got[i] = sched.synthesize(texts[i], "sky", max_new_tokens=budgets[i])

    threads = [threading.Thread(target=client, args=(i,)) for i in range(len(budgets))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    st = sched.stats()
    sched.close()
    assert st["completed"] == len(budgets) and st["failed"] == 0 and st["awaiting_codec"] == 0
    for i, (g, w) in enumerate(zip(got, want)):
        assert g is not None and g.shape == w.shape, (i, None if g is None else g.shape, w.shape)
        assert float(np.sqrt(np.mean((g - w) ** 2))) <= 1e-6, i


def test_scheduler_with_the_servers_sampling_defaults_runs_clean():
    """temp 0.5 / fast temp 0 / min_p 0.1 (server/settings.py:33-38): nothing to compare the audio with, but every request must
    finish with the right amount of finite audio, and the same text asked twice must not come out identical every time."""
    from smoltts_amd import SmolTTS
    from smoltts_amd.codec.synthetic import synthetic_mimi_state
    from smoltts_amd.config import GenerationSettings
    from smoltts_amd.server.scheduler import BatchScheduler
    from smoltts_amd.synthetic import named_config, synthetic_lm_state

    cfg = named_config("tiny")
    tts = SmolTTS(state=synthetic_lm_state(cfg, seed=21), config=cfg, mimi_state=synthetic_mimi_state(seed=5))
    gs = GenerationSettings(default_temp=0.5, default_fast_temp=0.0, min_p=0.1, max_new_tokens=12)
    sched = BatchScheduler(tts, max_batch=3, frames_per_tick=2, generation_settings=gs)
    n_req = 30
    out = [None] * n_req

    def client(i):
        r = sched.submit("the same words every time", "heart", stream=bool(i % 2), max_new_tokens=4 + i % 9)
        out[i] = np.concatenate(list(sched.iter_chunks(r)) or [np.zeros(0, np.float32)])

    threads = [threading.Thread(target=client, args=(i,)) for i in range(n_req)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    st = sched.stats()
    sched.close()
    assert st["completed"] == n_req and st["failed"] == 0
    for i, pcm in enumerate(out):
        n = 4 + i % 9
        assert pcm is not None and np.isfinite(pcm).all() and pcm.shape[0] % 1920 == 0
        if i % 2:
            assert 1920 <= pcm.shape[0] <= (n + 1) * 1920  # streams carry every generated frame (fewer only after <|im_end|>)
        else:
            assert pcm.shape[0] <= (n + 1) * 1920
    same_budget = [out[i] for i in range(1, n_req, 2) if 4 + i % 9 == 9]  # streams with the same text and budget
    assert len(same_budget) >= 2 and any(a.shape != b.shape or not np.array_equal(a, b) for a in same_budget for b in same_budget if a is not b)
