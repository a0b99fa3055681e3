"""Voice-clone prompt path end to end: recordings -> Mimi codes -> speaker prompt grid ->
generation conditioned on it, each stage against the CPU oracles."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_create_speaker_and_conditioned_generation():
    from oracle.lm_oracle import LMOracle, OracleLMConfig
    from oracle.mimi_oracle import MimiDecodeOracle, MimiEncodeOracle
    from smoltts_amd import SmolTTS
    from smoltts_amd.codec.synthetic import synthetic_mimi_encoder_state, synthetic_mimi_state, synthetic_pcm
    from smoltts_amd.config import GenerationSettings
    from smoltts_amd.synthetic import named_config, synthetic_lm_state

    cfg = named_config("tiny")
    state = synthetic_lm_state(cfg, seed=13)
    mst = {**synthetic_mimi_state(seed=4), **synthetic_mimi_encoder_state(seed=4)}
    tts = SmolTTS(state=state, config=cfg, mimi_state=mst)
    samples = [{"text": "first reference line", "audio": synthetic_pcm(1920 * 7 + 100, 1)},
               {"text": "and a second one", "audio": synthetic_pcm(1920 * 4, 2)[None, None]}]  # (1,1,L) as codec.encode takes it
    spk = tts.create_speaker(samples, system_prompt="You are a narrator.")

    # the same grid built from the oracle's codes (reference create_speaker, __init__.py:97-118)
    eorc = MimiEncodeOracle(mst, 8)
    pe, turns = tts.prompt_encoder, []
    turns.append(pe.encode_text_turn("system", "You are a narrator."))
    for s in samples:
        codes = eorc.encode(torch.from_numpy(np.asarray(s["audio"]).reshape(-1))[None, None])[0].numpy()
        assert codes.shape[0] == 8
        turns.append(pe.encode_text_turn("user", s["text"]))
        block = np.concatenate([codes[0:1] + 320, codes], axis=0)
        turns.append(np.concatenate([block, pe.tokenize_text("<|im_end|>\n")], axis=1))
    want = np.concatenate(turns, axis=1)
    assert spk.shape == want.shape == (9, want.shape[1]) and np.array_equal(spk, want)
    n_audio = 8 + 4
    assert int(((spk[0] >= 320) & (spk[0] <= 2367)).sum()) == n_audio
    assert np.array_equal(spk[1], np.where((spk[0] >= 320) & (spk[0] <= 2367), spk[0] - 320, 0))  # duplicated code 0

    # conditioned synthesis == oracle pipeline on the same prompt
    gs = GenerationSettings.greedy(max_new_tokens=7)
    pcm = tts("cloned voice", None, speaker=spk, generation_settings=gs)
    prompt = tts._get_prompt("cloned voice", "heart", spk)
    assert np.array_equal(prompt[:, : spk.shape[1]], spk)
    lorc = LMOracle(OracleLMConfig.from_dict(cfg.__dict__), state)
    grid = lorc.generate([torch.from_numpy(prompt)], max_frames=8, stop_on_eos=True)[0].as_tensor()
    keep = (grid[0] >= 320) & (grid[0] <= 2367)
    ref = MimiDecodeOracle(mst).decode(grid[1:, keep][None])[0, 0].numpy()
    assert pcm.shape == ref.shape
    assert float(np.sqrt(np.mean((pcm - ref) ** 2))) <= 1e-4

    with pytest.raises(ValueError, match="text"):
        tts.create_speaker([{"audio": np.zeros(10, np.float32)}])
    dec_only = SmolTTS(state=state, config=cfg, mimi_state=synthetic_mimi_state(seed=4))
    with pytest.raises(ValueError, match="encoder"):
        dec_only.create_speaker(samples)
