"""Capacity boundaries: the longest utterance the reference allows (max_new_tokens = 1024 -> 1025 frames, lm/generate.py:60,161)
through the façade, the streaming iterator and the scheduler; sequence-length errors are reported, not faults."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_full_length_utterance_everywhere():
    from smoltts_amd import SmolTTS
    from smoltts_amd.codec.synthetic import synthetic_mimi_state
    from smoltts_amd.config import GenerationSettings
    from smoltts_amd.engine import SmolttsError
    from smoltts_amd.server.scheduler import BatchScheduler
    from smoltts_amd.synthetic import named_config, synthetic_lm_state

    cfg = named_config("tiny")
    cfg.max_seq_len = 1100  # prompt (~30) + 1025 frames fit; the model's RoPE table has exactly this many rows
    tts = SmolTTS(state=synthetic_lm_state(cfg, seed=3), config=cfg, mimi_state=synthetic_mimi_state(seed=3))
    gs = GenerationSettings.greedy(max_new_tokens=1024)
    chunks = list(tts.stream("as long as it gets", "heart", generation_settings=gs))
    assert len(chunks) == 1025 and all(c.shape == (1920,) for c in chunks)
    full = np.concatenate(chunks)
    assert np.isfinite(full).all()
    sched = BatchScheduler(tts, max_batch=2, frames_per_tick=8, generation_settings=gs)
    got = np.concatenate(list(sched.iter_chunks(sched.submit("as long as it gets", "heart", stream=True))))
    assert got.shape == full.shape and float(np.sqrt(np.mean((got - full) ** 2))) <= 1e-6
    blocking = sched.synthesize("as long as it gets", "heart")
    assert blocking.shape[0] % 1920 == 0 and 0 < blocking.shape[0] <= full.shape[0]
    with pytest.raises(ValueError, match="max_seq_len"):
        sched.synthesize("x" * 200)  # 200 + 1024 frames do not fit 1100 positions
    sched.close()
    with pytest.raises((SmolttsError, ValueError)):
        tts("y" * 1200, "heart", generation_settings=GenerationSettings.greedy(max_new_tokens=4))  # prompt alone too long
