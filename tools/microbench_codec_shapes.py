"""Mimi decode cost per frame by call shape (utterances per pass x frames per chunk x utterance length): what a codec pass
of the serving scheduler costs next to the bench's 32 x 32 chunks."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from smoltts_amd.codec.synthetic import synthetic_mimi_state  # noqa: E402
from smoltts_amd.engine import MimiEngine, MimiSession  # noqa: E402


def main():
    eng = MimiEngine(synthetic_mimi_state(seed=0), num_codebooks=8, max_positions=2 * (1026 + 64))
    rng = np.random.default_rng(0)
    for batch, chunk, F in [(32, 32, 288), (32, 32, 32), (8, 64, 192), (8, 128, 256), (16, 128, 256), (16, 64, 256), (32, 64, 256), (4, 64, 256), (1, 64, 256)]:
        sess = MimiSession(eng, max_batch=batch, max_chunk_frames=chunk)
        codes = torch.from_numpy(rng.integers(0, 2048, size=(batch, F, 8)).astype(np.int32)).cuda()
        pcm = torch.empty(batch, F * 1920, dtype=torch.float32, device="cuda")
        best = 1e9
        for _ in range(3):
            sess.reset()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for f0 in range(0, F, chunk):
                sess.decode_chunk(codes, f0, min(chunk, F - f0), pcm)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t)
        print(f"batch {batch:3d} x chunk {chunk:3d}, {F:3d} frames each: {best * 1e3:7.2f} ms = {best * 1e6 / (batch * F):6.2f} us per frame", flush=True)
        sess.close()


if __name__ == "__main__":
    main()
