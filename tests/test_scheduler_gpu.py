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


def test_scheduler_stops_on_eos_like_the_facade():
    """<|im_end|> ends an utterance inside a tick: the stream delivers the frames up to and including the terminating one,
    the blocking answer drops it (non-semantic slow id), the slot is re-used, and everything equals the single-request façade."""
    import dataclasses

    from smoltts_amd import SmolTTS
    from smoltts_amd.codec.synthetic import synthetic_mimi_state
    from smoltts_amd.config import GenerationSettings
    from smoltts_amd.engine import LMEngine
    from smoltts_amd.generate import generate_blocking
    from smoltts_amd.server.scheduler import BatchScheduler
    from smoltts_amd.synthetic import named_config, synthetic_lm_state

    cfg = named_config("tiny")
    state = synthetic_lm_state(cfg, seed=21)
    tts = SmolTTS(state=state, config=cfg, mimi_state=synthetic_mimi_state(seed=5))
    gs = GenerationSettings.greedy(max_new_tokens=40)
    texts = ["first request", "second, longer request text", "3", "the fourth one"]
    # pick as <|im_end|> the slow id that the first utterance emits at frame 5 (synthetic weights never say 270 by themselves)
    grid = generate_blocking(tts.lm, tts._get_prompt(texts[0], "heart"), gs, audio_only=False)
    eos = int(grid[0, 0, 5])
    tts.token_config = dataclasses.replace(tts.token_config, im_end_id=eos)
    tts.lm = LMEngine(cfg, state, tts.token_config, tts.lm.numerics)
    want_stream = [np.concatenate(list(tts.stream(t, "heart", generation_settings=gs))) for t in texts]
    want_block = [tts(t, "heart", generation_settings=gs) for t in texts]
    assert want_stream[0].shape[0] <= 6 * 1920  # the first utterance really stops at (or before) its frame 5
    sched = BatchScheduler(tts, max_batch=2, frames_per_tick=4, generation_settings=gs)
    got_s, got_b = [None] * 4, [None] * 4

    def worker(i, stream):
        if stream:
            got_s[i] = np.concatenate(list(sched.iter_chunks(sched.submit(texts[i], "heart", stream=True))) or [np.zeros(0, np.float32)])
        else:
            got_b[i] = sched.synthesize(texts[i], "heart")

    threads = [threading.Thread(target=worker, args=(i, s)) for i in range(4) for s in (True, False)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    sched.close()
    for i in range(4):
        assert got_s[i] is not None and got_s[i].shape == want_stream[i].shape, (i, got_s[i].shape, want_stream[i].shape)
        assert float(np.sqrt(np.mean((got_s[i] - want_stream[i]) ** 2))) <= 1e-6 if got_s[i].size else True
        assert got_b[i] is not None and got_b[i].shape == want_block[i].shape, (i, got_b[i].shape, want_block[i].shape)
        assert float(np.sqrt(np.mean((got_b[i] - want_block[i]) ** 2))) <= 1e-6 if got_b[i].size else True


def test_scheduler_cancel_frees_the_slot_and_leaves_the_others_alone():
    """A client that abandons its stream (or its queued request) gives the slot back within a tick; whoever gets the slot
    next, and the requests decoding beside it, still receive exactly the single-request audio."""
    import time

    from smoltts_amd import SmolTTS
    from smoltts_amd.codec.synthetic import synthetic_mimi_state
    from smoltts_amd.config import GenerationSettings
    from smoltts_amd.server.scheduler import BatchScheduler
    from smoltts_amd.synthetic import named_config, synthetic_lm_state

    cfg = named_config("tiny")
    tts = SmolTTS(state=synthetic_lm_state(cfg, seed=21), config=cfg, mimi_state=synthetic_mimi_state(seed=5))
    gs = GenerationSettings.greedy(max_new_tokens=300)  # long enough that nobody finishes by themselves meanwhile
    want_long = np.concatenate(list(tts.stream("the neighbour keeps talking", "sky", generation_settings=gs)))
    assert want_long.shape[0] == 301 * 1920  # no <|im_end|> on the way: the victim below would run as long
    want_next = tts("takes over the abandoned slot", "nova", generation_settings=GenerationSettings.greedy(max_new_tokens=9))
    sched = BatchScheduler(tts, max_batch=2, frames_per_tick=2, generation_settings=gs)
    try:
        neighbour = sched.submit("the neighbour keeps talking", "sky", stream=True)
        victim = sched.submit("this client hangs up early", "heart", stream=True)
        queued = sched.submit("never gets a slot", "liam", stream=False)  # both slots are taken: it waits
        sched.cancel(queued)
        it = sched.iter_chunks(victim)
        first = next(it)
        assert first.shape[0] % 1920 == 0 and first.shape[0] > 0
        it.close()  # what a dropped HTTP connection does to the response generator
        assert victim.cancelled
        t0 = time.time()
        got_next = sched.synthesize("takes over the abandoned slot", "nova", max_new_tokens=9)  # needs the victim's slot
        assert time.time() - t0 < 30
        got_long = np.concatenate(list(sched.iter_chunks(neighbour)))
        assert got_next.shape == want_next.shape and float(np.sqrt(np.mean((got_next - want_next) ** 2))) <= 1e-6
        assert got_long.shape == want_long.shape and float(np.sqrt(np.mean((got_long - want_long) ** 2))) <= 1e-6
        assert list(sched.iter_chunks(queued)) == []  # ended without audio, without an error
        # the victim's queue ends too (a reader that comes back later does not block for ever)
        rest = list(sched.iter_chunks(victim))
        assert first.shape[0] + sum(c.shape[0] for c in rest) < 200 * 1920  # cut short, not played out
        st = sched.stats()
        assert st["completed"] == 2 and st["cancelled"] == 2 and st["failed"] == 0 and st["active"] == 0 and st["queued"] == 0
        assert st["frames_delivered"] >= (want_next.shape[0] + want_long.shape[0]) // 1920
    finally:
        sched.close()


def test_scheduler_splits_admissions_that_exceed_the_prefill_workspace_and_fails_loudly():
    """More prompt rows arriving at once than one prefill call can take are admitted over several calls, in order; an engine
    failure during admission answers every request — also the ones being admitted at that moment — and later submits raise."""
    from smoltts_amd import SmolTTS
    from smoltts_amd.codec.synthetic import synthetic_mimi_state
    from smoltts_amd.config import GenerationSettings
    from smoltts_amd.server.scheduler import BatchScheduler
    from smoltts_amd.synthetic import named_config, synthetic_lm_state

    cfg = named_config("tiny")
    tts = SmolTTS(state=synthetic_lm_state(cfg, seed=21), config=cfg, mimi_state=synthetic_mimi_state(seed=5))
    gs = GenerationSettings.greedy(max_new_tokens=5)
    texts = [f"request number {i}, some text" for i in range(8)]  # ~50 prompt columns each
    want = [tts(t, "sky", generation_settings=gs) for t in texts]
    sched = BatchScheduler(tts, max_batch=8, frames_per_tick=2, generation_settings=gs, max_prompt_rows=120, prefill_chunk=None)
    reqs = [sched.submit(t, "sky") for t in texts]  # 8 x ~50 rows against a 120-row workspace: at most two per call
    got = [np.concatenate(list(sched.iter_chunks(r)) or [np.zeros(0, np.float32)]) for r in reqs]
    for i, (g, w) in enumerate(zip(got, want)):
        assert g.shape == w.shape and float(np.sqrt(np.mean((g - w) ** 2))) <= 1e-6, i
    with pytest.raises(ValueError, match="workspace"):
        sched.synthesize("x" * 200)  # one prompt that no call can take: refused, the scheduler keeps serving
    assert sched.synthesize(texts[0], "sky").shape == want[0].shape

    def boom(*a, **k):
        raise RuntimeError("injected engine failure")

    sched.session.prefill = boom
    victims = [sched.submit(t, "sky") for t in texts[:3]]
    for r in victims:
        with pytest.raises(RuntimeError, match="injected engine failure"):
            list(sched.iter_chunks(r))
    sched._thread.join(timeout=30)
    with pytest.raises(RuntimeError, match="not running"):
        sched.submit("anyone there?")
    sched.close()


def test_scheduler_close_with_drain_finishes_what_it_accepted():
    from smoltts_amd import SmolTTS
    from smoltts_amd.codec.synthetic import synthetic_mimi_state
    from smoltts_amd.config import GenerationSettings
    from smoltts_amd.server.scheduler import BatchScheduler
    from smoltts_amd.synthetic import named_config, synthetic_lm_state

    cfg = named_config("tiny")
    tts = SmolTTS(state=synthetic_lm_state(cfg, seed=21), config=cfg, mimi_state=synthetic_mimi_state(seed=5))
    gs = GenerationSettings.greedy(max_new_tokens=20)
    want = tts("finish me first", "sky", generation_settings=gs)
    sched = BatchScheduler(tts, max_batch=2, frames_per_tick=2, generation_settings=gs)
    reqs = [sched.submit("finish me first", "sky", stream=bool(i % 2)) for i in range(5)]  # more than the slots: some still queued
    sched.close(drain=True)
    with pytest.raises(RuntimeError, match="not running"):
        sched.submit("too late")
    for i, r in enumerate(reqs):
        got = np.concatenate(list(sched.iter_chunks(r)))
        if not i % 2:
            assert got.shape == want.shape and float(np.sqrt(np.mean((got - want) ** 2))) <= 1e-6
        else:
            assert got.shape[0] == 21 * 1920
