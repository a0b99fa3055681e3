"""Prompt prefill beside the decode frames (smoltts_lm_park_slots / smoltts_lm_prefill_side / smoltts_lm_start_slots, the serving
loop's refill path since round 4) against the in-line deferred prefill it replaces (smoltts_lm_prefill_deferred): the same prompts
enter the same slots at the same frame boundaries, once in line and once with their KV rows computed on a second stream WHILE the
other slots decode -- every id of every slot must be the same, and the first tenants' ids must equal the CPU oracle's.
Reference loop being served: mlx_inference/src/smoltts_mlx/lm/generate.py:59-171 per request."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TEXTS = [("the first tenant", "heart"), ("a second one, speaking meanwhile", "nova"), ("arrives later", "sky"),
         ("and a fourth, with a much longer prompt than the others so that the side call outlasts a frame", "bella")]


def _setup(name):
    from smoltts_amd.config import TokenConfig
    from smoltts_amd.engine import LMEngine
    from smoltts_amd.prompt import PromptEncoder
    from smoltts_amd.synthetic import named_config, synthetic_lm_state
    from smoltts_amd.tokenizer import load_tokenizer

    cfg = named_config(name)
    tok = load_tokenizer()
    tc = TokenConfig.from_tokenizer(tok, cfg)
    state = synthetic_lm_state(cfg, seed=13)
    eng = LMEngine(cfg, state, tc)
    pe = PromptEncoder(tok, tc.semantic_start_id, cfg.num_codebooks, cfg.duplicate_code_0)
    return cfg, state, eng, [pe.build_prompt(t, v) for t, v in TEXTS]


def _run(eng, prompts, side: bool, F=10):
    from smoltts_amd.engine import LMSession

    s = LMSession(eng, max_batch=4, max_seq=256, max_rows=512, max_frames=40)
    frames = torch.cuda.current_stream()
    other = torch.cuda.Stream()
    s.prefill(prompts[:2], slots=[0, 1], stop_on_eos=False, defer_frame0=True)
    s.decode(3)
    if side:
        h = s.side_park(prompts[2:], [2, 3])          # frame stream
        s.decode(2)                                   # the frames the side call runs beside
        with torch.cuda.stream(other):
            s.side_run(h)                             # (waits on the host for the park, not for the two frames)
        s.side_start(h, stop_on_eos=False)            # frame stream again, after the host saw the side call finish
    else:
        s.decode(2)
        s.prefill(prompts[2:], slots=[2, 3], stop_on_eos=False, defer_frame0=True)
    s.decode(F)
    frames.synchronize()
    codes, n, done, margin = s.fetch()
    s.close()
    return codes, n


@pytest.mark.parametrize("name", ["tiny", "smoltts_byte_70m"])
def test_side_prefill_gives_the_ids_of_the_inline_prefill(name):
    cfg, state, eng, prompts = _setup(name)
    a, na = _run(eng, prompts, side=True)
    b, nb = _run(eng, prompts, side=False)
    assert np.array_equal(na, nb) and list(na[:2]) == [15, 15] and list(na[2:]) == [10, 10]
    for slot in range(4):
        assert np.array_equal(a[slot, :na[slot]], b[slot, :nb[slot]]), f"slot {slot}: ids differ between side and in-line prefill"
    # ... and they are the oracle's
    from oracle.lm_oracle import LMOracle, OracleLMConfig

    orc = LMOracle(OracleLMConfig.from_dict(cfg.__dict__), state)
    logs = orc.generate([torch.from_numpy(p) for p in prompts], max_frames=10, stop_on_eos=False)
    for slot in range(4):
        want = np.array(logs[slot].grid)
        assert np.array_equal(a[slot, :10], want[:10]), f"slot {slot}: ids differ from the oracle"
    eng.close()


def test_park_refuses_positions_outside_the_cache():
    from smoltts_amd.engine import LMSession, SmolttsError

    cfg, state, eng, prompts = _setup("tiny")
    s = LMSession(eng, max_batch=2, max_seq=24, max_rows=64, max_frames=4)
    with pytest.raises(SmolttsError):
        s.side_park([prompts[3]], [1])  # the prompt does not fit the session's cache
    s.close()
    eng.close()
