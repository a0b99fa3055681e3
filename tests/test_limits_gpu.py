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


def test_a_slot_stops_when_its_context_is_full():
    """A slot whose next token would land at position max_seq stops there (device stop rule) instead of decoding on without
    KV / RoPE rows; the frames it did emit are the oracle's, and a later tenant of the slot is unaffected."""
    import torch

    from oracle.lm_oracle import LMOracle, OracleLMConfig
    from smoltts_amd.config import TokenConfig
    from smoltts_amd.engine import LMEngine, LMSession
    from smoltts_amd.prompt import PromptEncoder
    from smoltts_amd.synthetic import named_config, synthetic_lm_state
    from smoltts_amd.tokenizer import load_tokenizer

    cfg = named_config("tiny")
    state = synthetic_lm_state(cfg, seed=5)
    tok = load_tokenizer()
    tc = TokenConfig.from_tokenizer(tok, cfg)
    pe = PromptEncoder(tok, tc.semantic_start_id, cfg.num_codebooks, cfg.duplicate_code_0)
    long_p, short_p = pe.build_prompt("a rather long prompt, as prompts go", "heart"), pe.build_prompt("hi", "sky")
    T = long_p.shape[1]
    max_seq = T + 6  # room for frame 0 (from the prompt) + 6 more tokens at positions T .. T+5
    eng = LMEngine(cfg, state, tc)
    sess = LMSession(eng, 2, max_seq=max_seq, max_rows=128, max_frames=32)
    sess.prefill([long_p, short_p], stop_on_eos=False)
    sess.decode(20)
    codes, n, done, _ = sess.fetch()
    assert n[0] == 7 and done[0] == 1, (n, done)                    # positions T..T+5 consumed, then the context is full
    assert n[1] == min(21, max_seq - short_p.shape[1] + 1) and n[1] > n[0]
    orc = LMOracle(OracleLMConfig.from_dict(cfg.__dict__), state)
    logs = orc.generate([torch.from_numpy(long_p), torch.from_numpy(short_p)], max_frames=21, stop_on_eos=False)
    for b in range(2):
        assert np.array_equal(codes[b, : n[b]], np.array(logs[b].grid)[: n[b]])
    # the slot is reusable: a new tenant starts from its own prompt
    sess.prefill([short_p], slots=[0], stop_on_eos=False)
    sess.decode(3)
    codes2, n2, _, _ = sess.fetch()
    assert n2[0] == 4 and np.array_equal(codes2[0, :4], np.array(logs[1].grid)[:4])
    sess.close()
