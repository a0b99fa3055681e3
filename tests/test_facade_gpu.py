"""The reference-shaped façade end to end on the GPU: checkpoint directory (both weight formats) ->
SmolTTS.__call__ / stream / generate_blocking, against the CPU oracles."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(tmp_path_factory):
    from oracle.lm_oracle import LMOracle, OracleLMConfig
    from oracle.mimi_oracle import MimiDecodeOracle
    from smoltts_amd import SmolTTS
    from smoltts_amd.checkpoint import save_checkpoint
    from smoltts_amd.codec.synthetic import synthetic_mimi_state
    from smoltts_amd.synthetic import named_config, synthetic_lm_state

    cfg = named_config("tiny")
    state = synthetic_lm_state(cfg, seed=11)
    mst = synthetic_mimi_state(seed=2)
    d = save_checkpoint(tmp_path_factory.mktemp("ckpt_st"), cfg, state, fmt="safetensors", mlx_layout=True)
    d2 = save_checkpoint(tmp_path_factory.mktemp("ckpt_pth"), cfg, state, fmt="pth")
    tts = SmolTTS(checkpoint_dir=str(d), mimi_state=mst)
    orc = LMOracle(OracleLMConfig.from_dict(cfg.__dict__), state)
    return cfg, state, mst, tts, orc, MimiDecodeOracle(mst), d2


def _rms(a):
    return float(np.sqrt(np.mean(np.square(a, dtype=np.float64))))


def test_call_matches_oracle_pipeline(setup):
    from smoltts_amd.config import GenerationSettings

    cfg, state, mst, tts, orc, morc, _ = setup
    gs = GenerationSettings.greedy(max_new_tokens=9)
    pcm = tts("Hello world!", "sky", generation_settings=gs)
    prompt = tts._get_prompt("Hello world!", "sky")
    log = orc.generate([torch.from_numpy(prompt)], max_frames=10, stop_on_eos=True)[0]
    grid = log.as_tensor()  # (9, F)
    keep = (grid[0] >= 320) & (grid[0] <= 2367)
    codes = grid[1:, keep]
    ref = morc.decode(codes[None])[0, 0].numpy()
    assert pcm.dtype == np.float32 and pcm.shape == ref.shape == (1920 * int(keep.sum()),)
    assert _rms(pcm - ref) <= 1e-4


def test_stream_chunks_concatenate_to_batch_decode(setup):
    from smoltts_amd.config import GenerationSettings

    cfg, state, mst, tts, orc, morc, _ = setup
    gs = GenerationSettings.greedy(max_new_tokens=5)
    chunks = list(tts.stream("streaming", "heart", generation_settings=gs))
    assert len(chunks) == 6 and all(c.shape == (1920,) and c.dtype == np.float32 for c in chunks)
    prompt = tts._get_prompt("streaming", "heart")
    grid = orc.generate([torch.from_numpy(prompt)], max_frames=6, stop_on_eos=True)[0].as_tensor()
    ref = morc.decode(grid[1:][None])[0, 0].numpy()  # stream decodes every frame's codes (reference __init__.py:88-92)
    assert _rms(np.concatenate(chunks) - ref) <= 1e-4


def test_stream_with_the_reference_upsample(setup):
    """``stream(..., reference_upsample=True)``: the chunks of the reference's own ``stream`` (every frame up-sampled alone,
    codec/mimi.py:77) -- the oracle's per-call form at one frame per call."""
    from smoltts_amd.config import GenerationSettings

    cfg, state, mst, tts, orc, morc, _ = setup
    gs = GenerationSettings.greedy(max_new_tokens=5)
    chunks = list(tts.stream("streaming", "heart", generation_settings=gs, reference_upsample=True))
    prompt = tts._get_prompt("streaming", "heart")
    grid = orc.generate([torch.from_numpy(prompt)], max_frames=6, stop_on_eos=True)[0].as_tensor()
    ref = morc.decode(grid[1:][None], upsample_call_frames=1)[0, 0].numpy()
    assert len(chunks) == 6 and _rms(np.concatenate(chunks) - ref) <= 1e-4
    assert _rms(np.concatenate(chunks) - morc.decode(grid[1:][None])[0, 0].numpy()) > 1e-3


def test_overlapped_stream_yields_the_same_chunks(setup):
    """``stream`` runs the codec step of frame f beside frame f + 1 (generate.stream_pcm): same chunks, bit for bit, as the
    one-stream loop, also when the utterance ends by <|im_end|> or by its frame budget."""
    from smoltts_amd.config import GenerationSettings

    cfg, state, mst, tts, orc, morc, _ = setup
    for n in (1, 2, 9):
        gs = GenerationSettings.greedy(max_new_tokens=n)
        a = list(tts.stream("two streams, one answer", "sky", generation_settings=gs, overlap=True))
        b = list(tts.stream("two streams, one answer", "sky", generation_settings=gs, overlap=False))
        assert len(a) == len(b) == n + 1
        assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_generate_blocking_and_pth_checkpoint(setup):
    from smoltts_amd import SmolTTS
    from smoltts_amd.config import GenerationSettings
    from smoltts_amd.generate import SingleBatchGenerator, generate_blocking

    cfg, state, mst, tts, orc, morc, d2 = setup
    tts2 = SmolTTS(checkpoint_dir=str(d2), mimi_state=mst)  # torch-side layout: model.pth, (n, d, 2048) head
    prompt = tts._get_prompt("same ids from both checkpoint formats", "nova")
    gs = GenerationSettings.greedy(max_new_tokens=7)
    a = generate_blocking(tts.lm, prompt[None], gs, audio_only=False)
    b = generate_blocking(tts2.lm, prompt, gs, audio_only=False)
    want = orc.generate([torch.from_numpy(prompt)], max_frames=8, stop_on_eos=False)[0].as_tensor().numpy()
    assert a.shape == (1, 9, 8) and np.array_equal(a, b) and np.array_equal(a[0], want)
    frames = list(SingleBatchGenerator(tts.lm, prompt, gs))
    assert len(frames) == 8 and [f.semantic_code for f in frames] == want[0].tolist()
    for f, col in zip(frames, want.T):
        sem = 320 <= col[0] <= 2367
        assert (f.audio_codes is not None) == sem
        if sem:
            assert f.audio_codes.shape == (1, 8, 1) and f.audio_codes[0, :, 0].tolist() == col[1:].tolist()
    # reference defaults (temp 0.7 / 0.7): sampled on the device, reproducible per seed
    s1 = generate_blocking(tts.lm, prompt, GenerationSettings(max_new_tokens=7, seed=5), audio_only=False)
    s2 = generate_blocking(tts.lm, prompt, GenerationSettings(max_new_tokens=7, seed=5), audio_only=False)
    s3 = generate_blocking(tts.lm, prompt, GenerationSettings(max_new_tokens=7, seed=6), audio_only=False)
    assert np.array_equal(s1, s2) and not np.array_equal(s1, s3) and not np.array_equal(s1, a)
    # slow sampled, depth greedy (the server's defaults: temp 0.5, fast temp 0.0)
    s4 = generate_blocking(tts.lm, prompt, GenerationSettings(default_temp=0.5, default_fast_temp=0.0, min_p=0.1, max_new_tokens=7, seed=1), audio_only=False)
    assert s4.shape == (1, 9, 8)


def test_call_reports_the_reference_timing_figures(setup, capsys):
    """lm/generate.py:187-214 prints prefill ms / tokens per s and frames per s / ms per frame / x realtime; here the same
    figures are kept in ``last_stats`` (and printed with ``verbose``)."""
    from smoltts_amd.config import GenerationSettings

    cfg, state, mst, tts, orc, morc, _ = setup
    tts.verbose = True
    try:
        pcm = tts("Hello world!", "sky", generation_settings=GenerationSettings.greedy(max_new_tokens=9))
    finally:
        tts.verbose = False
    st = tts.last_stats
    assert st["utterances"] == 1 and st["prompt_tokens"] == 24 and st["frames"] == 10  # 5 + 16 + 3 prompt columns
    assert st["prefill_ms"] > 0 and st["decode_s"] > 0 and st["frames_per_s"] > 0
    assert abs(st["realtime_x"] - st["frames_per_s"] / 12.5) < 1e-9
    assert st["codec_frames"] == pcm.shape[0] // 1920 and st["codec_ms"] > 0
    out = capsys.readouterr().out
    assert "Prompt: 24 tokens" in out and "x realtime" in out and "to PCM" in out
