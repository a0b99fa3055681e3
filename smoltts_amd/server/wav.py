"""PCM -> WAV framing (mlx_inference/src/smoltts_mlx/io/wav.py:4-37): 44-byte RIFF header, PCM16 mono,
samples = (pcm * 32767).astype(int16)."""
import numpy as np


def pcm_to_wav_bytes(pcm_data: np.ndarray, sample_rate: int = 24000) -> bytes:
    pcm_data = np.asarray(pcm_data).flatten()
    n = len(pcm_data) * 2
    header = b"".join([
        b"RIFF", (n + 36).to_bytes(4, "little"), b"WAVE", b"fmt ", (16).to_bytes(4, "little"),
        (1).to_bytes(2, "little"), (1).to_bytes(2, "little"), int(sample_rate).to_bytes(4, "little"),
        (int(sample_rate) * 2).to_bytes(4, "little"), (2).to_bytes(2, "little"), (16).to_bytes(2, "little"),
        b"data", n.to_bytes(4, "little"),
    ])
    return header + (pcm_data * 32767).astype(np.int16).tobytes()
