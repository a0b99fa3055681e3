"""Mimi-decode-only harness for rocprofv3 --pmc runs: B=32 slots, chunks of 32 frames (bench.py's Mimi step)."""
import sys

import torch

sys.path.insert(0, ".")
from smoltts_amd.codec.synthetic import synthetic_mimi_state  # noqa: E402
from smoltts_amd.engine import MimiEngine, MimiSession, load_library  # noqa: E402

load_library()
eng = MimiEngine(synthetic_mimi_state(seed=0), 8, max_positions=400)
sess = MimiSession(eng, max_batch=32, max_chunk_frames=32)
codes = torch.randint(0, 2048, (32, 96, 8), dtype=torch.int32).cuda()
pcm = torch.empty(32, 96 * 1920, device="cuda")
sess.reset()
for f0 in (0, 32, 64):
    sess.decode_chunk(codes, f0, 32, pcm)
    torch.cuda.synchronize()
print("ok", float(pcm.abs().mean()))
