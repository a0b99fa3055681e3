"""CPU suite: the oracle against the committed golden vectors (which were produced against the
reference's own RQTransformer.forward / against transformers.MimiModel), plus its self-consistency."""
import numpy as np
import pytest
import torch


def _lm(cfgname, seed):
    from oracle.lm_oracle import LMOracle, OracleLMConfig
    from smoltts_amd.synthetic import named_config, state_fingerprint, synthetic_lm_state

    cfg = named_config(cfgname)
    state = synthetic_lm_state(cfg, seed=seed)
    return cfg, state, LMOracle(OracleLMConfig.from_dict(cfg.__dict__), state), state_fingerprint(state)


@pytest.mark.parametrize("name", ["tiny", "tiny_nodup", "tiny_proj", "70m"])
def test_lm_oracle_reproduces_goldens(name, golden_dir):
    g = np.load(golden_dir / f"lm_{name}.npz")
    cfg, state, orc, fp = _lm(str(g["config_name"]), int(g["seed"]))
    assert abs(fp - float(g["fingerprint"])) <= 1e-6 * abs(fp)
    n = len(g["texts"])
    logs = orc.generate([torch.from_numpy(g[f"prompt_{b}"]) for b in range(n)], max_frames=int(g["frames"]), stop_on_eos=False)
    for b in range(n):
        assert np.array_equal(logs[b].as_tensor().numpy(), g[f"grid_{b}"])
        # the reference forward's own logits at three positions (teacher-forced on the golden grid)
        full = torch.cat([torch.from_numpy(g[f"prompt_{b}"]).long(), torch.from_numpy(g[f"grid_{b}"]).long()], dim=1)
        tok, cb = orc.teacher_forced(full)
        rows = g[f"ref_rows_{b}"].tolist()
        assert np.allclose(tok[rows].numpy(), g[f"ref_token_logits_{b}"], atol=5e-5)
        assert np.allclose(cb[rows][:, :, :64].numpy(), g[f"ref_codebook_logits_{b}"], atol=2e-6)


def test_lm_oracle_batch_invariance_and_eos():
    """Batched decode == one-by-one decode; stop rule: the <|im_end|> frame is emitted, then nothing."""
    from oracle.lm_oracle import LMOracle, OracleLMConfig
    from smoltts_amd.prompt import PromptEncoder
    from smoltts_amd.synthetic import named_config, synthetic_lm_state
    from smoltts_amd.tokenizer import load_tokenizer

    cfg = named_config("tiny")
    state = synthetic_lm_state(cfg, seed=5)
    pe = PromptEncoder(load_tokenizer(), 320)
    prompts = [torch.from_numpy(pe.build_prompt(t, "heart")) for t in ("abc", "a longer second prompt", "x")]
    ocfg = OracleLMConfig.from_dict(cfg.__dict__)
    batched = LMOracle(ocfg, state).generate(prompts, max_frames=6, stop_on_eos=False)
    for b, p in enumerate(prompts):
        single = LMOracle(ocfg, state).generate([p], max_frames=6, stop_on_eos=False)[0]
        assert single.grid == batched[b].grid
    eos = batched[1].grid[2][0]
    ocfg2 = OracleLMConfig.from_dict({**cfg.__dict__, "im_end_id": eos})
    logs = LMOracle(ocfg2, state).generate(prompts, max_frames=6, stop_on_eos=True)
    first = next(f for f in range(6) if batched[1].grid[f][0] == eos)
    assert len(logs[1].grid) == first + 1 and logs[1].grid == batched[1].grid[: first + 1]


def test_lm_oracle_mlx_mode_differs_only_by_documented_quirks():
    from oracle.lm_oracle import LMOracle, OracleLMConfig, rope_table
    from smoltts_amd.synthetic import named_config, synthetic_lm_state

    cfg = named_config("tiny")
    state = synthetic_lm_state(cfg, seed=1)
    ocfg = OracleLMConfig.from_dict(cfg.__dict__)
    a, b = LMOracle(ocfg, state, "torch", True), LMOracle(ocfg, state, "mlx", False)
    cols = torch.tensor([[72] + [0] * 8, [320 + 5] + [5, 1, 2, 3, 4, 5, 6, 7], [400] + [0, 9, 9, 9, 9, 9, 9, 9]])
    ea, eb = a.embed(cols), b.embed(cols)
    assert torch.equal(ea[0], eb[0]) and torch.equal(ea[1], eb[1])  # text row / ordinary audio row agree
    assert not torch.equal(ea[2], eb[2])  # code0 == 0 on a semantic row: torch zeroes the code sum, MLX keeps it
    t_bf, t_ex = rope_table(64, 64, 1e5, True), rope_table(64, 64, 1e5, False)
    assert float((t_bf - t_ex).abs().max()) < 4e-3 and not torch.equal(t_bf, t_ex)


def test_mimi_oracle_matches_hf_goldens_and_live_model(golden_dir):
    from oracle.mimi_oracle import MimiDecodeOracle
    from smoltts_amd.codec.synthetic import synthetic_mimi_state

    g = np.load(golden_dir / "mimi_hf.npz")
    st = synthetic_mimi_state(seed=int(g["seed"]))
    fp = float(sum(float(v.double().abs().sum()) for v in st.values()))
    assert abs(fp - float(g["fingerprint"])) <= 1e-9 * fp
    codes = torch.from_numpy(g["codes"]).long()
    out = MimiDecodeOracle(st, window=250).decode(codes).numpy()
    assert out.shape == g["pcm"].shape
    assert float(np.sqrt(np.mean((out - g["pcm"]) ** 2))) < 1e-6
    # window = 0 (MLX behaviour) is identical while the context is shorter than 250 positions
    assert np.array_equal(MimiDecodeOracle(st, window=0).decode(codes).numpy(), out)
    transformers = pytest.importorskip("transformers")
    m = transformers.MimiModel(transformers.MimiConfig()).eval()
    m.load_state_dict(st, strict=False)
    codes2 = torch.randint(0, 2048, (1, 8, 9), generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        ref = m.decode(codes2)[0].numpy()
    assert float(np.sqrt(np.mean((MimiDecodeOracle(st, window=250).decode(codes2).numpy() - ref) ** 2))) < 1e-6


def test_mimi_oracle_is_causal():
    from oracle.mimi_oracle import MimiDecodeOracle
    from smoltts_amd.codec.synthetic import synthetic_mimi_state

    orc = MimiDecodeOracle(synthetic_mimi_state(seed=1))
    codes = torch.randint(0, 2048, (1, 8, 7), generator=torch.Generator().manual_seed(0))
    full, prefix = orc.decode(codes), orc.decode(codes[:, :, :3])
    assert float((full[..., : 3 * 1920] - prefix).abs().max()) < 1e-5


def test_mimi_oracle_per_call_upsample_is_the_reference_stream():
    """``decode(codes, upsample_call_frames=c)``: what a stream of ``decode_step`` calls of c frames yields in the reference
    (codec/mimi.py:73-77: ``self.upsample`` sees the call's frames only; conv.py:273-282: full transposed conv, k - stride = 2 rows
    trimmed on the right).  Checked against that definition written out per frame, and against the properties that follow from it."""
    import torch.nn.functional as F

    from oracle.mimi_oracle import MimiDecodeOracle
    from smoltts_amd.codec.synthetic import synthetic_mimi_state

    st = synthetic_mimi_state(seed=2)
    orc = MimiDecodeOracle(st)
    codes = torch.randint(0, 2048, (2, 8, 6), generator=torch.Generator().manual_seed(5))
    batch = orc.decode(codes)
    # one call over the whole utterance (or more): the batch decode, bit for bit
    assert torch.equal(orc.decode(codes, upsample_call_frames=6), batch) and torch.equal(orc.decode(codes, upsample_call_frames=64), batch)
    # c = 1, from the definition: rows (2f, 2f + 1) = e[f] * w[:, 0], e[f] * w[:, 1]; taps 2, 3 fall into the trimmed rows
    e = orc.rvq_decode(codes.long())  # B, 512, F
    w = st["upsample.conv.weight"]  # 512, 1, 4
    rows = []
    for f in range(6):
        y = F.conv_transpose1d(e[:, :, f:f + 1], w, None, stride=2, groups=512)  # B, 512, 4
        rows.append(y[:, :, :2])
        assert torch.allclose(y[:, :, 0], e[:, :, f] * w[:, 0, 0]) and torch.allclose(y[:, :, 1], e[:, :, f] * w[:, 0, 1])
    up1 = torch.cat(rows, dim=2)
    ref1 = orc.seanet(orc.transformer(up1.transpose(1, 2)).transpose(1, 2))
    one = orc.decode(codes, upsample_call_frames=1)
    assert torch.allclose(one, ref1, atol=1e-6)
    # the first call has nothing to lose; later calls differ from the batch decode (the reference's defect)
    two = orc.decode(codes, upsample_call_frames=2)
    assert float((two[..., : 2 * 1920] - batch[..., : 2 * 1920]).abs().max()) < 1e-5
    assert float((two[..., 2 * 1920:] - batch[..., 2 * 1920:]).abs().max()) > 1e-3
    assert float((one[..., 1920:] - batch[..., 1920:]).abs().max()) > 1e-3
