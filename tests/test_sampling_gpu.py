"""Sampling parity is statistical (the reference's MLX generator is not reproducible): the device
sampler must draw from softmax(logits / temp), restricted by min_p, and be a pure function of its key."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sample(logits_row, n, temp, min_p, seed, frame_base=0, step=0):
    from smoltts_amd import engine as E

    lib = E.load_library()
    logits = logits_row[None].repeat(n, 1).contiguous().cuda()
    ids = torch.empty(n, dtype=torch.int32, device="cuda")
    E.check(lib.smoltts_k_sample(E.dptr(logits), n, logits.shape[1], logits.stride(0), temp, min_p, seed, frame_base, step,
                                 E.dptr(ids), E.current_stream_ptr()), "smoltts_k_sample")
    return ids.cpu().numpy()


def _chi2_ok(counts, probs, n):
    keep = probs * n >= 5
    exp = probs[keep] * n
    chi2 = float(((counts[keep] - exp) ** 2 / exp).sum())
    dof = int(keep.sum()) - 1
    return chi2 < dof + 6 * math.sqrt(2 * dof), chi2, dof  # ~6 sigma


@pytest.mark.parametrize("temp,min_p,mode", [(1.0, None, "reference"), (0.7, None, "intended"), (0.5, 0.1, "intended"), (0.5, 0.1, "reference")])
def test_sampler_follows_softmax(temp, min_p, mode):
    """Both readings of ``min_p``: "reference" = what lm/utils/samplers.py:22-28 computes (the threshold is the token's own
    log-probability + log(min_p), so nothing is removed: categorical over logits / temp, the drop-in default); "intended" =
    keep p >= min_p * p_max.  The device gets ``GenerationSettings.effective_min_p``."""
    from smoltts_amd.config import GenerationSettings

    g = torch.Generator().manual_seed(3)
    V = 2368
    row = torch.randn(V, generator=g) * 2.0
    n = 60000
    eff = GenerationSettings(default_temp=temp, min_p=min_p, min_p_mode=mode).effective_min_p
    assert eff == (min_p if (mode == "intended" and min_p) else 0.0)
    ids = _sample(row, n, temp, eff, seed=12345)
    z = (row - row.max()) / temp
    if eff > 0:
        z = torch.where(z >= math.log(eff), z, torch.full_like(z, float("-inf")))
        assert set(np.unique(ids)).issubset(set(torch.nonzero(torch.isfinite(z)).flatten().tolist()))
    elif min_p:  # reference mode with the server's min_p = 0.1: tokens below the "intended" cut must still be drawn
        below = (z < math.log(min_p)).numpy()
        assert below[ids].sum() > 0.5 * n * float(torch.softmax(z.double(), 0)[torch.from_numpy(below)].sum())
    probs = torch.softmax(z.double(), dim=0).numpy()
    counts = np.bincount(ids, minlength=V).astype(np.float64)
    ok, chi2, dof = _chi2_ok(counts, probs, n)
    assert ok, f"chi2 {chi2:.1f} for {dof} dof"


def test_sampler_is_a_pure_function_of_its_key():
    row = torch.randn(2048, generator=torch.Generator().manual_seed(1))
    a = _sample(row, 256, 0.7, 0.0, seed=7, frame_base=3, step=2)
    assert np.array_equal(a, _sample(row, 256, 0.7, 0.0, seed=7, frame_base=3, step=2))
    for kw in (dict(seed=8, frame_base=3, step=2), dict(seed=7, frame_base=4, step=2), dict(seed=7, frame_base=3, step=3)):
        assert not np.array_equal(a, _sample(row, 256, 0.7, 0.0, **kw))
    # frame_base + row is the frame counter: shifting the base shifts the draws along the rows
    b = _sample(row, 256, 0.7, 0.0, seed=7, frame_base=4, step=2)
    assert len(np.unique(a)) > 20 and not np.array_equal(a[1:], b[:-1])  # the row index is part of the key too


def test_session_sampling_modes_and_greedy_limit():
    from smoltts_amd.config import TokenConfig
    from smoltts_amd.engine import LMEngine, LMSession
    from smoltts_amd.prompt import PromptEncoder
    from smoltts_amd.synthetic import named_config, synthetic_lm_state
    from smoltts_amd.tokenizer import load_tokenizer

    cfg = named_config("tiny")
    tok = load_tokenizer()
    eng = LMEngine(cfg, synthetic_lm_state(cfg, seed=2), TokenConfig.from_tokenizer(tok, cfg))
    pe = PromptEncoder(tok, 320)
    prompts = [pe.build_prompt("sampling modes", "heart"), pe.build_prompt("b", "sky")]

    def run(**kw):
        s = LMSession(eng, 2, max_seq=128, max_rows=128, max_frames=8)
        s.set_sampling(**kw)
        s.prefill(prompts, stop_on_eos=False)
        s.decode(7)
        out = s.fetch()[0].copy()
        s.close()
        return out

    greedy = run()
    assert np.array_equal(greedy, run(temp=1e-8, fast_temp=1e-8, seed=3))       # temperature -> 0 is greedy (the depth logits are O(1e-2))
    slow_only = run(temp=1.5, fast_temp=0.0, seed=3)
    assert np.array_equal(slow_only, run(temp=1.5, fast_temp=0.0, seed=3))     # replayable
    assert not np.array_equal(slow_only[:, :, 0], greedy[:, :, 0])            # slow ids are sampled
    both = run(temp=1.5, fast_temp=1.5, seed=3)
    assert np.array_equal(both[:, 0, 0], slow_only[:, 0, 0])                  # same key => same first slow draw
    assert not np.array_equal(both[:, :, 1:], slow_only[:, :, 1:])


def test_a_slots_next_tenant_draws_its_own_numbers():
    """Same seed, same slot, same prompt, same frame numbers: the second tenant of a slot must still not repeat the first
    one's samples (a serving session restarts slots all day), while a fresh session replays the first tenant exactly."""
    from smoltts_amd.config import TokenConfig
    from smoltts_amd.engine import LMEngine, LMSession
    from smoltts_amd.prompt import PromptEncoder
    from smoltts_amd.synthetic import named_config, synthetic_lm_state
    from smoltts_amd.tokenizer import load_tokenizer

    cfg = named_config("tiny")
    tok = load_tokenizer(None, cfg.codebook_size)
    tc = TokenConfig.from_tokenizer(tok, cfg)
    eng = LMEngine(cfg, synthetic_lm_state(cfg, seed=2), tc)
    prompt = PromptEncoder.from_config(tok, cfg, tc).encode_text_turn("user", "the same request twice")

    def tenant(sess, deferred):
        sess.prefill([prompt], slots=[0], stop_on_eos=False, defer_frame0=deferred)
        sess.decode(12)
        codes, n, _, _ = sess.fetch()
        return codes[0, : int(n[0])].copy()

    for deferred in (False, True):
        a = LMSession(eng, 2, max_seq=128, max_rows=64, max_frames=16)
        a.set_sampling(temp=1.5, fast_temp=1.5, seed=11)
        first, second = tenant(a, deferred), tenant(a, deferred)
        a.close()
        b = LMSession(eng, 2, max_seq=128, max_rows=64, max_frames=16)
        b.set_sampling(temp=1.5, fast_temp=1.5, seed=11)
        replay = tenant(b, deferred)
        b.close()
        assert np.array_equal(first, replay)
        assert first.shape == second.shape and not np.array_equal(first, second)
