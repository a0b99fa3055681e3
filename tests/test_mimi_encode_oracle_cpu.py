"""The encode-half oracle (voice-clone prompts) against vectors produced by
``transformers.MimiModel.encode`` (tests/golden/make_mimi_goldens.py)."""
import numpy as np
import torch


def _state(seed):
    from smoltts_amd.codec.synthetic import synthetic_mimi_encoder_state, synthetic_mimi_state

    return {**synthetic_mimi_state(seed=seed), **synthetic_mimi_encoder_state(seed=seed)}


def test_encode_oracle_reproduces_third_party_vectors(golden_dir):
    from oracle.mimi_oracle import MimiEncodeOracle
    from smoltts_amd.codec.synthetic import synthetic_pcm

    g = np.load(golden_dir / "mimi_enc_hf.npz")
    st = _state(int(g["seed"]))
    for L in g["lengths"].tolist():
        pcm = torch.from_numpy(synthetic_pcm(L, int(g["pcm_seed"])))[None, None]
        whole = L % 1920 == 0
        for extra_right in ((False, True) if whole else (True,)):  # ragged lengths: only the transformers convention is pinned
            orc = MimiEncodeOracle(st, 8, window=250, extra_right=extra_right)
            emb = orc.embeddings(pcm)
            assert float((emb - torch.from_numpy(g[f"emb_{L}"])).abs().max()) < 2e-5
            codes, gap = orc.rvq_encode(emb, return_margin=True)
            assert np.array_equal(codes.numpy(), g[f"codes_{L}"]) and gap > 1e-3


def test_encode_oracle_properties():
    from oracle.mimi_oracle import MimiDecodeOracle, MimiEncodeOracle
    from smoltts_amd.codec.synthetic import synthetic_pcm

    st = _state(5)
    orc = MimiEncodeOracle(st, 8)
    pcm = torch.from_numpy(synthetic_pcm(1920 * 6, 2))[None, None]
    codes = orc.encode(pcm)
    assert codes.shape == (1, 8, 6) and int(codes.min()) >= 0 and int(codes.max()) < 2048
    # causal: the codes of a whole-frame prefix are a prefix of the codes
    assert torch.equal(orc.encode(pcm[..., : 1920 * 4]), codes[..., :4])
    # ragged length: ceil(L / 1920) frames, and the reference's all-left padding == prepending zeros
    L = 1920 * 3 + 500
    assert orc.encode(pcm[..., :L]).shape[-1] == 4
    # residual quantisation really reduces the residual: decoding more codebooks gets closer to the latent
    emb = orc.embeddings(pcm)
    p = "quantizer.acoustic_residual_vector_quantizer."
    r = torch.nn.functional.conv1d(emb, st[p + "input_proj.weight"]).transpose(1, 2)[0]
    norms = [float(r.norm())]
    for i in range(7):
        cb = orc._codebook(p + f"layers.{i}.codebook.")
        r = r - cb[codes[0, 1 + i]]
        norms.append(float(r.norm()))
    assert all(b < a for a, b in zip(norms, norms[1:]))
    # batch of two == one by one
    two = torch.cat([pcm, torch.from_numpy(synthetic_pcm(1920 * 6, 9))[None, None]])
    assert torch.equal(orc.encode(two)[0], codes[0])
    assert MimiDecodeOracle(st).decode(codes).shape == (1, 1, 1920 * 6)
