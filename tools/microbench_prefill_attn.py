"""Prefill attention timing: packed short prompts (the bench's 3539 rows) and one long prompt (voice-clone speaker)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from smoltts_amd import engine, ops  # noqa: E402

engine.load_library()
Hq, Hkv = 12, 4


def run(lengths, cache_len, label):
    slots = len(lengths)
    kc = torch.randn(slots, Hkv, cache_len, 64, device="cuda")
    vc = torch.randn(slots, Hkv, cache_len, 64, device="cuda")
    pos = torch.cat([torch.arange(n) for n in lengths]).int().cuda()
    slot = torch.cat([torch.full((n,), s) for s, n in enumerate(lengths)]).int().cuda()
    q = torch.randn(pos.numel(), Hq * 64, device="cuda")
    for _ in range(3):
        ops.attention(q, kc, vc, pos, slot, Hq)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ops.attention(q, kc, vc, pos, slot, Hq)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 20 * 1e6
    pairs = sum(n * (n + 1) // 2 for n in lengths)
    print(f"{label:34s} rows={pos.numel():5d} (row, key) pairs={pairs:9d}: {us:8.1f} us  {4 * 64 * Hq * pairs / us / 1e6:6.1f} TFLOP/s")


g = torch.Generator().manual_seed(2)
run([int(x) for x in torch.randint(58, 166, (32,), generator=g)], 176, "32 packed prompts (bench)")
run([500], 512, "one prompt of 500")
run([1000], 1024, "one prompt of 1000")
run([2000], 2048, "one prompt of 2000")
