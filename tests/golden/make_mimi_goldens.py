"""Generate the committed Mimi golden vectors (runs wherever `transformers` is installed).

The Mimi arithmetic of the reference lives in third-party code: the MLX port under
mlx_inference/src/smoltts_mlx/codec/ (not importable: no `mlx`) mirrors `transformers.MimiModel`
(used by the reference's data_pipeline/utils/codec.py:4,19).  This script loads our seeded
synthetic decoder-side weights into `transformers.MimiModel`, decodes seeded codes with it and
stores (seed, codes, pcm) as data.  tests/test_oracle_cpu.py checks oracle/mimi_oracle.py against
both the live third-party model and these vectors; tests/test_round2_gpu.py and tests/test_mimi_gpu.py decode them on
the HIP engine (also replicated to the benchmark's chunk size).

Usage: python tests/golden/make_mimi_goldens.py
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from smoltts_amd.codec.synthetic import synthetic_mimi_encoder_state, synthetic_mimi_state, synthetic_pcm  # noqa: E402


def long_vector():
    """A longer utterance (30 frames = 60 transformer positions, several SEANet halo refills in streaming decode)."""
    from transformers import MimiConfig, MimiModel

    seed, B, F = 5, 1, 30
    st = synthetic_mimi_state(seed=seed)
    m = MimiModel(MimiConfig()).eval()
    res = m.load_state_dict(st, strict=False)
    assert not res.unexpected_keys
    codes = torch.randint(0, 2048, (B, 8, F), generator=torch.Generator().manual_seed(12))
    with torch.no_grad():
        pcm = m.decode(codes)[0]
    fp = float(sum(float(v.double().abs().sum()) for v in st.values()))
    np.savez_compressed(Path(__file__).resolve().parent / "mimi_hf_long.npz", seed=seed, codes=codes.numpy().astype(np.int32),
                        pcm=pcm.numpy().astype(np.float32), fingerprint=fp)
    print("wrote mimi_hf_long.npz", pcm.shape, float(pcm.pow(2).mean().sqrt()))


def main():
    from transformers import MimiConfig, MimiModel

    if "--only-long" in sys.argv:
        return long_vector()
    long_vector()
    seed, B, F = 3, 2, 6
    st = synthetic_mimi_state(seed=seed)
    m = MimiModel(MimiConfig()).eval()
    res = m.load_state_dict(st, strict=False)
    assert not res.unexpected_keys
    g = torch.Generator().manual_seed(11)
    codes = torch.randint(0, 2048, (B, 8, F), generator=g)
    with torch.no_grad():
        pcm = m.decode(codes)[0]
    fp = float(sum(float(v.double().abs().sum()) for v in st.values()))
    np.savez_compressed(Path(__file__).resolve().parent / "mimi_hf.npz", seed=seed, codes=codes.numpy().astype(np.int32),
                        pcm=pcm.numpy().astype(np.float32), fingerprint=fp)
    print("wrote mimi_hf.npz", pcm.shape, float(pcm.pow(2).mean().sqrt()))

    # ---- encode half (voice-clone prompts): MimiModel.encode on seeded encoder-side weights.
    # 7680 samples = 4 whole frames (both padding conventions agree); 6460 is ragged: transformers pads the
    # stride-alignment extra on the right, the reference's MLX conv on the left (oracle ``extra_right``).
    st = {**synthetic_mimi_state(seed=seed), **synthetic_mimi_encoder_state(seed=seed)}
    m = MimiModel(MimiConfig()).eval()
    res = m.load_state_dict(st, strict=False)
    assert not res.unexpected_keys
    save = {"seed": seed, "pcm_seed": 1, "lengths": np.array([7680, 6460])}
    for L in (7680, 6460):
        x = torch.from_numpy(synthetic_pcm(L, 1))[None, None]
        with torch.no_grad():
            save[f"codes_{L}"] = m.encode(x, num_quantizers=8).audio_codes.numpy().astype(np.int32)
            e = m.encoder(x)
            e = m.encoder_transformer(e.transpose(1, 2))[0].transpose(1, 2)
            save[f"emb_{L}"] = m.downsample(e).numpy().astype(np.float32)
    np.savez_compressed(Path(__file__).resolve().parent / "mimi_enc_hf.npz", **save)
    print("wrote mimi_enc_hf.npz", {k: getattr(v, "shape", v) for k, v in save.items()})


if __name__ == "__main__":
    main()
