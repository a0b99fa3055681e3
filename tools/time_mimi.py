"""Time of one Mimi chunk decode (default 32 slots x 32 frames = 1024 frames), HIP events, best of 3 streams.
argv: [slots] [frames] [chunks per stream, default 7]: the best chunk (early in the stream) and the LAST chunk of the stream
(the codec transformer attends to everything before it: window 0) are reported.  CODEC_PRODUCTS=3: SMOLTTS_MIMI_OPT_PRODUCTS."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from smoltts_amd.codec.synthetic import synthetic_mimi_state  # noqa: E402
from smoltts_amd.engine import MimiEngine, MimiSession  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
F = int(sys.argv[2]) if len(sys.argv) > 2 else 32
NCH = int(sys.argv[3]) if len(sys.argv) > 3 else 7
eng = MimiEngine(synthetic_mimi_state(seed=0), 8, window=0, max_positions=2 * F * (NCH + 1) + 16)
sess = MimiSession(eng, max_batch=B, max_chunk_frames=F, products=int(os.environ.get("CODEC_PRODUCTS", 6)))
codes = torch.randint(0, 2048, (B, F * NCH, 8), dtype=torch.int32, device="cuda")
pcm = torch.zeros(B, F * NCH * 1920, device="cuda")
best, last = 1e9, 1e9
for rep in range(3):
    sess.reset()
    for i in range(NCH):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        sess.decode_chunk(codes, i * F, F, pcm, code_offset=0)
        b.record()
        torch.cuda.synchronize()
        if i:
            best = min(best, a.elapsed_time(b))
        if i == NCH - 1:
            last = min(last, a.elapsed_time(b))
print(f"Mimi chunk decode {B} slots x {F} frames: {best:.3f} ms ({best * 1e3 / (B * F):.2f} us per frame)")
print(f"chunk {NCH - 1} of the stream (transformer positions {2 * F * (NCH - 1)}..{2 * F * NCH - 1}): {last:.3f} ms")
