"""Chunked prompt prefill (BASELINE config 5): chunks fill the KV cache only, the last chunk emits frame 0, decode
frames of the other slots run in between — ids must equal the oracle's one-utterance-at-a-time generation."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(cfgname, seed, weight_format="bf16"):
    from oracle.lm_oracle import LMOracle, OracleLMConfig
    from smoltts_amd.config import NumericsMode, TokenConfig
    from smoltts_amd.engine import LMEngine
    from smoltts_amd.packing import fp8_reference_state
    from smoltts_amd.prompt import PromptEncoder
    from smoltts_amd.synthetic import named_config, synthetic_lm_state
    from smoltts_amd.tokenizer import load_tokenizer

    cfg = named_config(cfgname)
    state = synthetic_lm_state(cfg, seed=seed)
    tok = load_tokenizer()
    eng = LMEngine(cfg, state, TokenConfig.from_tokenizer(tok, cfg), NumericsMode.torch_reference(), weight_format=weight_format)
    ocfg, ostate = fp8_reference_state(cfg, state) if weight_format == "fp8" else (cfg, state)
    orc = LMOracle(OracleLMConfig.from_dict(ocfg.__dict__), ostate)
    return cfg, eng, orc, PromptEncoder(tok, 320, cfg.num_codebooks, cfg.duplicate_code_0)


def _prompts(pe, lengths, seed=3):
    rng = np.random.default_rng(seed)
    return [pe.build_prompt("".join(chr(int(c)) for c in rng.integers(32, 127, size=n)), "nova") for n in lengths]


@pytest.mark.parametrize("cfgname,chunk,wf", [("tiny", 32, "bf16"), ("tiny", 7, "bf16"), ("smoltts_byte_70m", 128, "fp8")])
def test_chunked_equals_one_shot_and_oracle(cfgname, chunk, wf):
    from smoltts_amd.engine import LMSession

    cfg, eng, orc, pe = _setup(cfgname, 31, wf)
    prompts = _prompts(pe, [1, 40, 150, 290])  # T = 25 .. 314: zero to many chunks
    frames = 8
    a = LMSession(eng, max_batch=4, max_seq=400, max_rows=700, max_frames=frames)
    a.prefill(prompts, stop_on_eos=False)
    a.decode(frames - 1)
    one_shot = a.fetch()[0]
    b = LMSession(eng, max_batch=4, max_seq=400, max_rows=700, max_frames=frames)
    calls = []
    b.prefill_chunked(prompts, stop_on_eos=False, chunk=chunk, between=lambda: calls.append(1))
    assert len(calls) == max((p.shape[1] - 1) // chunk for p in prompts)
    b.decode(frames - 1)
    codes, n, done, margin = b.fetch()
    logs = orc.generate([torch.from_numpy(p) for p in prompts], max_frames=frames, stop_on_eos=False)
    for i in range(4):
        want = logs[i].as_tensor().numpy()
        assert n[i] == frames
        assert np.array_equal(codes[i, :frames].T, want), f"slot {i} (T={prompts[i].shape[1]}): chunked prefill differs from the oracle"
        assert np.array_equal(one_shot[i, :frames].T, want)
    a.close(); b.close(); eng.close()


def test_decode_runs_between_chunks_of_other_slots():
    """Slots 0,1 are speaking; slots 2,3 (long prompts) enter in chunks with decode frames in between; later slot 1 is
    re-used the same way.  Every slot's ids == its own oracle generation."""
    from smoltts_amd.engine import LMSession

    cfg, eng, orc, pe = _setup("tiny", 8)
    P = _prompts(pe, [20, 35, 200, 260, 180], seed=9)
    frames = 40
    s = LMSession(eng, max_batch=4, max_seq=400, max_rows=600, max_frames=frames)
    s.prefill(P[:2], slots=[0, 1], stop_on_eos=False)
    ticks = []

    def tick():
        s.decode(2)
        ticks.append(2)

    s.prefill_chunked(P[2:4], slots=[2, 3], stop_on_eos=False, chunk=64, between=tick)
    s.decode(3)
    codes, n, done, _ = s.fetch()
    t1 = sum(ticks)
    assert n.tolist() == [1 + t1 + 3, 1 + t1 + 3, 4, 4]
    # restart slot 1 with a chunked prompt while 0, 2, 3 keep going
    first_life_of_1 = codes[1, : n[1]].copy()
    s.prefill_chunked([P[4]], slots=[1], stop_on_eos=False, chunk=50, between=tick)
    s.decode(2)
    codes, n, done, _ = s.fetch()
    t2 = sum(ticks) - t1
    assert n.tolist() == [1 + t1 + 3 + t2 + 2, 3, 4 + t2 + 2, 4 + t2 + 2]
    logs = orc.generate([torch.from_numpy(p) for p in P], max_frames=frames, stop_on_eos=False)
    for slot, u in ((0, 0), (2, 2), (3, 3), (1, 4)):
        want = logs[u].as_tensor().numpy()[:, : n[slot]]
        assert np.array_equal(codes[slot, : n[slot]].T, want), f"slot {slot}"
    assert np.array_equal(first_life_of_1.T, logs[1].as_tensor().numpy()[:, : first_life_of_1.shape[0]])
    s.close(); eng.close()


@pytest.mark.parametrize("chunk", [None, 40])
def test_deferred_frame0_gives_the_same_ids(chunk):
    """smoltts_lm_prefill_deferred: KV fill only, frame 0 from the next decode frame — same ids as prefill + decode, also
    when other slots are in the middle of their utterances."""
    from smoltts_amd.engine import LMSession

    cfg, eng, orc, pe = _setup("tiny", 17)
    P = _prompts(pe, [30, 5, 120, 77], seed=4)
    frames = 20
    s = LMSession(eng, max_batch=4, max_seq=300, max_rows=400, max_frames=frames)
    s.prefill(P[:2], slots=[0, 1], stop_on_eos=False)          # slots 0, 1: ordinary start (frame 0 now)
    s.decode(5)
    if chunk:
        s.prefill_chunked(P[2:], slots=[2, 3], stop_on_eos=False, chunk=chunk, defer_frame0=True)
    else:
        s.prefill(P[2:], slots=[2, 3], stop_on_eos=False, defer_frame0=True)
    codes, n, done, _ = s.fetch()
    assert n.tolist() == [6, 6, 0, 0] and done.tolist()[2:] == [0, 0]
    s.decode(7)
    codes, n, done, _ = s.fetch()
    assert n.tolist() == [13, 13, 7, 7]
    logs = orc.generate([torch.from_numpy(p) for p in P], max_frames=frames, stop_on_eos=False)
    for slot in range(4):
        assert np.array_equal(codes[slot, : n[slot]].T, logs[slot].as_tensor().numpy()[:, : n[slot]]), f"slot {slot}"
    s.close(); eng.close()
