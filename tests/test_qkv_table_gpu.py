"""Depth layer-0 q | k | v out of the engine's derived table (smoltts_engine_build_fast_qkv) against the wqkv GEMM launch it
replaces, and the frame's slow token / last depth code picked inside the commit kernel.

The table holds what the decode path's own GEMM computes for every fast-embedding row (lm/generate.py:134-140: the input of depth
step i + 1 is fast_embeddings(code_i + offset_i), so layer 0's RMSNorm -> wqkv of it is a function of the table row alone);
with it a frame has 7 launches fewer.  Same values => the two launch sequences must emit the same ids; parity against the
reference itself is what the golden-grid tests check on the default (table) path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _prompts(cfg, tok, tc, texts):
    from smoltts_amd.prompt import PromptEncoder

    pe = PromptEncoder(tok, tc.semantic_start_id, cfg.num_codebooks, cfg.duplicate_code_0)
    return [pe.build_prompt(t, v) for t, v in texts]


TEXTS = [("the quick brown fox", "heart"), ("jumps over", "nova"), ("a lazy dog, twice; and then once more for the table", "sky"),
         ("0123456789", "bella"), ("z", "liam")]


@pytest.mark.parametrize("name,fmt", [("tiny", "bf16"), ("tiny_nodup", "bf16"), ("tiny_proj", "bf16"), ("smoltts_byte_70m", "bf16"), ("tiny", "fp8")])
def test_table_path_emits_the_ids_of_the_gemm_path(name, fmt):
    from smoltts_amd.config import TokenConfig
    from smoltts_amd.engine import LMEngine, LMSession
    from smoltts_amd.synthetic import named_config, synthetic_lm_state
    from smoltts_amd.tokenizer import load_tokenizer

    cfg = named_config(name)
    tok = load_tokenizer()
    tc = TokenConfig.from_tokenizer(tok, cfg)
    eng = LMEngine(cfg, synthetic_lm_state(cfg, seed=5), tc, weight_format=fmt)
    assert eng.fast_qkv is not None  # built by default
    prompts = _prompts(cfg, tok, tc, TEXTS)
    F = 14
    out = {}
    for on, picks in ((True, True), (False, False), (True, False), (False, True)):
        s = LMSession(eng, max_batch=len(prompts), max_seq=256, max_rows=512, max_frames=F)
        s.use_qkv_table(on)
        s.use_commit_picks(picks)
        s.prefill(prompts, stop_on_eos=False)
        s.decode(F - 1)
        codes, n, _, margin = s.fetch()
        assert (n == F).all()
        out[on, picks] = (codes[:, :F].copy(), margin.copy(), s.margin_at.cpu().numpy().copy())
        s.close()
    base = out[False, False]  # round 2's launch sequence: every GEMM and every pick a launch of its own
    for key, got in out.items():
        assert np.array_equal(got[0], base[0]), f"ids differ between launch structures {key} and (False, False)"
        # the smallest top-2 gaps agree too (the RoPE of a gathered row may differ from the GEMM epilogue's by one rounding)
        assert np.allclose(got[1], base[1], rtol=1e-3, atol=1e-6)
    assert np.array_equal(out[False, True][2], base[2])  # picking in the commit kernel moves no gap and no (frame, step) record
    eng.close()


def test_engine_without_the_table_still_decodes():
    from smoltts_amd.config import TokenConfig
    from smoltts_amd.engine import LMEngine, LMSession
    from smoltts_amd.synthetic import named_config, synthetic_lm_state
    from smoltts_amd.tokenizer import load_tokenizer

    cfg = named_config("tiny")
    tok = load_tokenizer()
    tc = TokenConfig.from_tokenizer(tok, cfg)
    st = synthetic_lm_state(cfg, seed=5)
    prompts = _prompts(cfg, tok, tc, TEXTS[:2])
    grids = []
    for table in (False, True):
        eng = LMEngine(cfg, st, tc, fast_qkv_table=table)
        assert (eng.fast_qkv is not None) == table
        s = LMSession(eng, max_batch=2, max_seq=128, max_rows=128, max_frames=8)
        s.prefill(prompts, stop_on_eos=False)
        s.decode(7)
        grids.append(s.fetch()[0][:, :8].copy())
        s.close(); eng.close()
    assert np.array_equal(grids[0], grids[1])


def test_sampled_frames_are_the_same_with_and_without_the_table():
    """Sampling keys are a function of (seed, slot, tenant, frame, step, column): picking the slow token and the last code in
    the commit kernel and the others beside a table gather must draw the same ids as separate launches would."""
    from smoltts_amd.config import TokenConfig
    from smoltts_amd.engine import LMEngine, LMSession
    from smoltts_amd.synthetic import named_config, synthetic_lm_state
    from smoltts_amd.tokenizer import load_tokenizer

    cfg = named_config("tiny")
    tok = load_tokenizer()
    tc = TokenConfig.from_tokenizer(tok, cfg)
    eng = LMEngine(cfg, synthetic_lm_state(cfg, seed=9), tc)
    prompts = _prompts(cfg, tok, tc, TEXTS[:3])
    got = []
    for on in (True, False):
        s = LMSession(eng, max_batch=3, max_seq=128, max_rows=256, max_frames=10)
        s.use_qkv_table(on)
        s.use_commit_picks(on)
        s.set_sampling(temp=0.8, fast_temp=0.6, min_p=0.05, seed=1234)
        s.prefill(prompts, stop_on_eos=False)
        s.decode(9)
        got.append(s.fetch()[0][:, :10].copy())
        s.close()
    assert np.array_equal(got[0], got[1])
    assert len({tuple(got[0][b].ravel()) for b in range(3)}) == 3  # (not a constant grid)
    eng.close()
