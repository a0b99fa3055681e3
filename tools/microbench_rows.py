"""Time the many-row GEMM on the SEANet shapes of one 1024-frame chunk (bottleneck experiments)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from smoltts_amd import engine, ops  # noqa: E402

engine.load_library()
shapes = [("convT4 128->64 s4", 491520, 256, 256, True), ("convT3 256->128 s5", 98304, 512, 640, True), ("res c3 64->32 k3", 1966080, 192, 32, False),
          ("res c1 32->64", 1966080, 32, 64, False), ("fc1 512->2048", 2048, 512, 2048, False)]
for name, M, K, N, raw in shapes:
    x = torch.randn(M, K, device="cuda")
    w = ops.pack_weight(torch.randn(N, K) / K ** 0.5, fp32=True)
    out = torch.empty(M, N, device="cuda")
    rawb = torch.empty(M, N, device="cuda") if raw else None
    for _ in range(2):
        ops.linear(x, w, N, w_fp32=True, out=out, elu_out=raw, raw_out=rawb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ops.linear(x, w, N, w_fp32=True, out=out, elu_out=raw, raw_out=rawb)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 5 * 1e6
    print(f"{name:22s} M={M:8d} K={K:4d} N={N:4d}: {us:8.1f} us  {2 * M * K * N / us / 1e6:6.1f} TFLOP/s  {(M * K + M * N * (2 if raw else 1)) * 4 / us / 1e3:7.1f} GB/s")
